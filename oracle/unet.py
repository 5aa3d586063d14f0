"""ORACLE (test infrastructure) — CQTDiff+ denoiser UNet, functional CPU restatement.

Follows /root/reference/networks/cqtdiff+.py:
  forward wiring            :730-845
  ResnetBlock               :382-493  (only the attention-free path; attention is
                                       disabled by conf/network/cqtdiff+.yaml:23)
  BiasFreeGroupNorm         :137-163
  RFF_MLP_Block             :167-211
  UpDownResample ('cubic')  :510-580
  Conv2d ("same", no bias)  :66-88
It consumes a ``state_dict`` with the reference's key names
and is differentiable through torch autograd (the checker for the HIP VJP).
Pinned against the imported reference by tests/golden (G6, G7).
"""
import math

import torch
import torch.nn.functional as F

RSQRT2 = 1.0 / math.sqrt(2.0)
CUBIC = [-0.01171875, -0.03515625, 0.11328125, 0.43359375,
         0.43359375, 0.11328125, -0.03515625, -0.01171875]


def group_norm_nomean(x, gamma, groups=8, eps=1e-7):
    """x / (unbiased std over the group + eps) * gamma; the mean is NOT removed from x (:147-163)."""
    B, C, Fq, T = x.shape
    xg = x.reshape(B, groups, -1)
    std = xg.std(dim=-1, keepdim=True)
    return (xg / (std + eps)).reshape(B, C, Fq, T) * gamma


def conv_same(x, w, dil_f=1):
    kh, kw = w.shape[-2:]
    pad = (dil_f * (kh - 1) // 2, (kw - 1) // 2)
    return F.conv2d(x, w, padding=pad, dilation=(dil_f, 1))


def resample_down(x):
    """time axis /2: reflect-pad 3, 8-tap FIR, stride 2 (:557-572)."""
    B, C, Fq, T = x.shape
    h = torch.tensor(CUBIC, dtype=x.dtype, device=x.device).view(1, 1, 8)
    xp = F.pad(x.reshape(B * C * Fq, 1, T), (3, 3), mode="reflect")
    return F.conv1d(xp, h, stride=2).reshape(B, C, Fq, -1)


def resample_up(x):
    """time axis x2: reflect-pad 2, transposed 8-tap FIR stride 2, crop 7 (:559-574)."""
    B, C, Fq, T = x.shape
    h = torch.tensor(CUBIC, dtype=x.dtype, device=x.device).view(1, 1, 8)
    xp = F.pad(x.reshape(B * C * Fq, 1, T), (2, 2), mode="reflect")
    return F.conv_transpose1d(xp, h, stride=2, padding=7).reshape(B, C, Fq, -1)


def linear(x, w, b):
    return x @ w.t() + b


def embedding(sd, cnoise):
    """RFF + 3-layer ReLU MLP, ReLU after the last layer too (:184-211). cnoise [B,1] -> [B,emb]."""
    table = 2 * math.pi * cnoise * sd["embedding.RFF_freq"]
    h = torch.cat([torch.sin(table), torch.cos(table)], dim=1)
    for i in range(3):
        h = torch.relu(linear(h, sd[f"embedding.MLP.{i}.weight"], sd[f"embedding.MLP.{i}.bias"]))
    return h


def resnet_block(sd, p, x_in, emb, num_dils, proj_after=False):
    """(:452-493). p = key prefix, e.g. 'downs.0.2.'"""
    x = x_in
    if p + "proj_in.weight" in sd:
        x = conv_same(x, sd[p + "proj_in.weight"])
    for d in range(num_dils):
        w = sd[p + f"H.{d}.weight"]
        dil = 2 ** d if w.shape[-2] > 1 else 1
        x0 = x
        h = group_norm_nomean(x, sd[p + f"norm.{d}.gamma"])
        gam = linear(emb, sd[p + f"affine.{d}.weight"], sd[p + f"affine.{d}.bias"])
        gate = linear(emb, sd[p + f"gate.{d}.weight"], sd[p + f"gate.{d}.bias"])
        h = h * (gam[:, :, None, None] + 1)
        x = (x0 + conv_same(F.gelu(h), w, dil) * gate[:, :, None, None]) * RSQRT2
    if proj_after and (p + "proj_out.weight") in sd:
        x = conv_same(x, sd[p + "proj_out.weight"])
    res = x_in
    if p + "res_conv.weight" in sd:
        res = conv_same(x_in, sd[p + "res_conv.weight"])
    return (x + res) * RSQRT2


def unet_body(sd, cfg, C_list, emb):
    """Everything between CQT.fwd and CQT.bwd (:746-839).

    C_list: list of numocts real tensors [B,2,binsoct,T_j], index 0 = LOWEST octave.
    Returns list of numocts real tensors [B,2,binsoct,T_j], same order.
    """
    nocts = cfg["num_octs"]
    bpo = cfg["bins_per_oct"]
    num_dils = cfg["num_dils"]
    hs = []
    X = pyr = None
    for i in range(nocts):
        C = C_list[nocts - 1 - i]
        C2 = resnet_block(sd, f"downs.{i}.0.", C, emb, 1)
        if i == 0:
            X = C2
            pyr = resample_down(C)
        elif i < nocts - 1:
            pyr = torch.cat((resample_down(C), resample_down(pyr)), dim=2)
            X = torch.cat((C2, X), dim=2)
        else:
            pyr = torch.cat((C, pyr), dim=2)
            X = torch.cat((C2, X), dim=2)
        X = resnet_block(sd, f"downs.{i}.2.", X, emb, num_dils[i])
        hs.append(X)
        if i < nocts - 1:
            X = resample_down(X)
        X = (X + conv_same(pyr, sd[f"downs.{i}.1.weight"])) * RSQRT2
    X = resnet_block(sd, "middle.0.1.", X, emb, num_dils[-1])
    Xout = resnet_block(sd, "middle.0.0.", X, emb, 1, proj_after=True)
    outs = [None] * nocts
    for i in range(nocts):
        j = nocts - 1 - i
        X = torch.cat((X, hs.pop()), dim=1)
        X = resnet_block(sd, f"ups.{i}.1.", X, emb, num_dils[j])
        Xout = (Xout + resnet_block(sd, f"ups.{i}.0.", X, emb, 1, proj_after=True)) * RSQRT2
        X = X[:, :, bpo:, :]
        outs[i] = Xout[:, :, :bpo, :]
        Xout = Xout[:, :, bpo:, :]
        if j > 0:
            X = resample_up(X)
            Xout = resample_up(Xout)
    return outs


def unet_forward(sd, cfg, cqt, x, cnoise):
    """net(x[B,L], cnoise[B,1]) -> [B,L]  (:730-845)."""
    emb = embedding(sd, cnoise)
    X_list = cqt.fwd(x.unsqueeze(1))
    C_list = [torch.view_as_real(c.squeeze(1)).permute(0, 3, 1, 2).contiguous() for c in X_list]
    outs = unet_body(sd, cfg, C_list, emb)
    O_list = [torch.view_as_complex(o.permute(0, 2, 3, 1).contiguous()).unsqueeze(1) for o in outs]
    y = cqt.bwd(O_list).squeeze(1)
    return y[:, : x.shape[-1]]
