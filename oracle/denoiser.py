"""ORACLE (test infrastructure, never imported by babe_amd/) — CPU restatement of the reference's denoiser pre-pass.

Follows /root/reference/networks/denoiser.py (MultiStage_denoise: two-stage STFT-domain U-Net, :232-321) and
testing/denoise_and_bwe_tester.py:109-165 (segmented application, STFT 1024/256, Hamming cross-fade).  Written as plain
functions over a reference-named state_dict; pinned by tests/golden/denoiser*.npz, which tests/golden/make_golden.py
produced by importing the reference itself on weights from `init_state_dict` below.

Tensor convention of the reference: X[B, 2 (re, im), T frames, F bins].
"""
import math

import torch
import torch.nn.functional as F

NS = [64, 64, 64, 128, 128, 256, 512]        # denoiser.py:243 (hard-coded widths)


def _conv_same_reflect(x, w, b):
    """nn.Conv2d(padding='same', padding_mode='reflect'), odd kernels (denoiser.py:40-46,74-78)."""
    kh, kw = w.shape[2:]
    if kh > 1 or kw > 1:
        x = F.pad(x, ((kw - 1) // 2, kw // 2, (kh - 1) // 2, kh // 2), mode="reflect")
    return F.conv2d(x, w, b)


def _dense_block(sd, pre, n, x):
    """DenseBlock.forward (denoiser.py:50-59): newest features are concatenated in FRONT."""
    x_ = F.elu(_conv_same_reflect(x, sd[f"{pre}.H.0.0.weight"], sd[f"{pre}.H.0.0.bias"]))
    for i in range(1, n):
        x = torch.cat((x_, x), 1)
        x_ = F.elu(_conv_same_reflect(x, sd[f"{pre}.H.{i}.0.weight"], sd[f"{pre}.H.{i}.0.bias"]))
    return x_


def _i_block(sd, pre, n, x):
    """I_Block.forward (denoiser.py:342-346): dense block + 1x1 projection of the input."""
    return _dense_block(sd, f"{pre}.tfc", n, x) + F.conv2d(x, sd[f"{pre}.conv2d_res.weight"], sd[f"{pre}.conv2d_res.bias"])


def _e_block(sd, pre, n, x):
    """E_Block.forward (denoiser.py:367-372): I_Block, then 4x4 stride-2 conv on a 2-sample reflect pad + ELU."""
    x = _i_block(sd, f"{pre}.i_block", n, x)
    d = F.conv2d(F.pad(x, (2, 2, 2, 2), mode="reflect"), sd[f"{pre}.conv2d_2.0.weight"], sd[f"{pre}.conv2d_2.0.bias"], stride=2)
    return F.elu(d), x


def _crop_to(a, ref_shape):
    """CropAddBlock / CropConcatBlock cropping of their first argument (denoiser.py:413-449)."""
    dh = (a.shape[2] - ref_shape[2]) // 2
    dw = (a.shape[3] - ref_shape[3]) // 2
    return a[:, :, dh:dh + ref_shape[2], dw:dw + ref_shape[3]]


def _d_block(sd, pre, n, x, bridge):
    """D_Block.forward (denoiser.py:397-410)."""
    up = F.elu(F.conv_transpose2d(x, sd[f"{pre}.tconv_1.0.weight"], sd[f"{pre}.tconv_1.0.bias"], stride=2))
    x2 = x.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)            # nn.Upsample(scale 2, nearest)
    if x2.shape[-1] != up.shape[-1]:                                           # always true: 2n vs 2n+2 (as written)
        x2 = F.conv2d(x2, sd[f"{pre}.projection.weight"], sd[f"{pre}.projection.bias"])
    y = _crop_to(up, x2.shape) + x2
    y = torch.cat((_crop_to(y, bridge.shape), bridge), 1)
    return _i_block(sd, f"{pre}.i_block", n, y)


def _encoder(sd, pre, depth, n, x):
    skips = []
    for i in range(depth):
        x, s = _e_block(sd, f"{pre}.eblocks.{i}", n, x)
        skips.append(s)
    return _i_block(sd, f"{pre}.i_block", n, x), skips


def _decoder(sd, pre, depth, n, x, skips):
    for i in range(depth - 1, -1, -1):
        x = _d_block(sd, f"{pre}.dblocks.{i}", n, x, skips[i])
    return x


def denoiser_forward(sd, cfg, X):
    """MultiStage_denoise.forward (denoiser.py:275-321).  cfg: depth, num_tfc, num_stages, use_SAM, use_fencoding.
    Returns (pred_stage_2, pred_stage_1) for two stages, pred_stage_1 otherwise."""
    depth, n = cfg["depth"], cfg["num_tfc"]
    xin = X
    if cfg["use_fencoding"]:
        emb = sd["freq_encoding.fembeddings"]                                  # [F, 10]
        e = emb.t()[None, :, None, :].expand(X.shape[0], 10, X.shape[2], emb.shape[0])
        xin = torch.cat((X, e), 1)
    x = F.elu(_conv_same_reflect(xin, sd["conv2d_1.0.weight"], sd["conv2d_1.0.bias"]))
    x, skips = _encoder(sd, "encoder_s1", depth, n, x)
    feats1 = _decoder(sd, "decoder_s1", depth, n, x, skips)
    if cfg["num_stages"] <= 1:
        return _conv_same_reflect(feats1, sd["finalblock.conv2.weight"], sd["finalblock.conv2.bias"])
    # SAM (denoiser.py:117-132)
    x1 = _conv_same_reflect(feats1, sd["sam_1.conv1.weight"], sd["sam_1.conv1.bias"])
    pred1 = _conv_same_reflect(feats1, sd["sam_1.conv2.weight"], sd["sam_1.conv2.bias"]) + X
    M = torch.sigmoid(_conv_same_reflect(pred1, sd["sam_1.conv3.weight"], sd["sam_1.conv3.bias"]))
    fout = x1 * M + feats1
    x = F.elu(_conv_same_reflect(xin, sd["conv2d_2.0.weight"], sd["conv2d_2.0.bias"]))
    x = torch.cat((x, fout if cfg["use_SAM"] else feats1), 1)
    x, skips = _encoder(sd, "encoder_s2", depth, n, x)
    feats2 = _decoder(sd, "decoder_s2", depth, n, x, skips)
    pred2 = _conv_same_reflect(feats2, sd["finalblock.conv2.weight"], sd["finalblock.conv2.bias"])
    return pred2, pred1


# ------------------------------------------------------------------------------------------------------------------
def param_shapes(cfg):
    """Reference state_dict names and shapes of MultiStage_denoise(cfg) (checked against the reference module in
    tests/golden/make_golden.py)."""
    depth, n = cfg["depth"], cfg["num_tfc"]
    nin = 12 if cfg["use_fencoding"] else 2
    out = {}
    if cfg["use_fencoding"]:
        out["freq_encoding.fembeddings"] = (cfg["f_dim"], 10)

    def conv(name, co, ci, kh, kw):
        out[f"{name}.weight"] = (co, ci, kh, kw)
        out[f"{name}.bias"] = (co,)

    def iblock(pre, n0, nn_):
        for i in range(n):
            conv(f"{pre}.tfc.H.{i}.0", nn_, n0 + i * nn_, 3, 3)
        conv(f"{pre}.conv2d_res", nn_, n0, 1, 1)

    def encoder(pre, n0):
        for i in range(depth):
            iblock(f"{pre}.eblocks.{i}.i_block", n0 if i == 0 else NS[i], NS[i])
            conv(f"{pre}.eblocks.{i}.conv2d_2.0", NS[i + 1], NS[i], 4, 4)
        iblock(f"{pre}.i_block", NS[depth], NS[depth])

    def decoder(pre):
        for i in range(depth):
            out[f"{pre}.dblocks.{i}.tconv_1.0.weight"] = (NS[i + 1], NS[i], 4, 4)      # ConvTranspose2d: [in, out, kh, kw]
            out[f"{pre}.dblocks.{i}.tconv_1.0.bias"] = (NS[i],)
            conv(f"{pre}.dblocks.{i}.projection", NS[i], NS[i + 1], 1, 1)
            iblock(f"{pre}.dblocks.{i}.i_block", 2 * NS[i], NS[i])

    conv("conv2d_1.0", NS[0], nin, 7, 7)
    encoder("encoder_s1", NS[0])
    decoder("decoder_s1")
    conv("finalblock.conv2", 2, NS[0], 3, 3)
    if cfg["num_stages"] > 1:
        conv("sam_1.conv1", NS[0], NS[0], 3, 3)
        conv("sam_1.conv2", 2, NS[0], 3, 3)
        conv("sam_1.conv3", NS[0], 2, 3, 3)
        conv("conv2d_2.0", NS[0], nin, 7, 7)
        encoder("encoder_s2", 2 * NS[0])
        decoder("decoder_s2")
    return out


def freq_embeddings(f_dim):
    """AddFreqEncoding.__init__ (denoiser.py:141-157): cos(pi n), cos(2^k pi n), k = 1..9, n = bin/(f_dim-1)."""
    pi = torch.acos(torch.zeros(1)).item() * 2
    nn_ = torch.arange(start=0, end=f_dim) / (f_dim - 1)
    cols = [torch.cos(pi * nn_)] + [torch.cos(2 ** k * pi * nn_) for k in range(1, 10)]
    return torch.stack(cols, -1)


def init_state_dict(cfg, seed=0):
    """Deterministic synthetic weights (no pretrained checkpoint is available): uniform with variance 1/fan_in per
    tensor, each tensor drawn from its own generator seeded by (seed, position in param_shapes)."""
    sd = {}
    for idx, (name, shp) in enumerate(param_shapes(cfg).items()):
        if name == "freq_encoding.fembeddings":
            sd[name] = freq_embeddings(cfg["f_dim"])
            continue
        g = torch.Generator().manual_seed(seed * 100003 + idx)
        if name.endswith(".bias"):
            sd[name] = (torch.rand(shp, generator=g) * 2 - 1) * 0.1
        else:
            fan_in = shp[1] * shp[2] * shp[3]
            if ".tconv_1." in name:
                fan_in = shp[0] * 4          # each output of the 4x4 stride-2 transposed conv sees 2x2 taps
            b = math.sqrt(3.0 / fan_in)
            sd[name] = (torch.rand(shp, generator=g) * 2 - 1) * b
    return sd


# ------------------------------------------------------------------------------------------------------------------
def apply_denoiser_model(sd, cfg, x, win_size=1024, hop_size=256):
    """denoise_and_bwe_tester.py:146-165: zero-pad by one window, STFT (periodic Hamming, center=False),
    network (first output if two stages), inverse STFT, crop."""
    window = torch.hamming_window(window_length=win_size)
    x = torch.cat((x, torch.zeros(x.shape[0], win_size)), -1)
    X = torch.stft(x, win_size, hop_length=hop_size, window=window, center=False, return_complex=True)
    X = torch.view_as_real(X).permute(0, 3, 2, 1)
    pred = denoiser_forward(sd, cfg, X)
    if cfg["num_stages"] > 1:
        pred = pred[0]
    pred = torch.view_as_complex(pred.permute(0, 3, 2, 1).contiguous())
    y = torch.istft(pred, win_size, hop_length=hop_size, window=window, center=False, return_complex=False)
    return y[..., 0:x.shape[-1]]


def apply_denoiser(sd, cfg, x, segment_size, win_size=1024, hop_size=256, overlapsize=1024, model=None):
    """denoise_and_bwe_tester.py:109-142: fixed-size segments hopping by segment_size - 1024, cross-faded with the two
    halves of a 2048-point periodic Hamming window; the last (zero-padded) segment is not faded out."""
    model = model or (lambda seg: apply_denoiser_model(sd, cfg, seg, win_size, hop_size))
    n = x.shape[-1]
    window = torch.hamming_window(window_length=2 * overlapsize)
    wl, wr = window[:overlapsize], window[overlapsize:]
    out = torch.zeros_like(x)
    pointer = 0
    while True:
        if pointer + segment_size < n:
            y = model(x[:, pointer:pointer + segment_size])
            parts = []
            if pointer == 0:
                parts = [y[:, :segment_size - overlapsize], y[:, segment_size - overlapsize:segment_size] * wr]
            else:
                parts = [y[:, :overlapsize] * wl, y[:, overlapsize:segment_size - overlapsize],
                         y[:, segment_size - overlapsize:segment_size] * wr]
            out[:, pointer:pointer + segment_size] += torch.cat(parts, -1)
            pointer += segment_size - overlapsize
        else:
            seg = x[:, pointer:]
            nl = seg.shape[-1]
            seg = torch.cat((seg, torch.zeros(seg.shape[0], segment_size - nl)), -1)
            y = model(seg)
            if pointer != 0:
                y = torch.cat((y[:, :overlapsize] * wl, y[:, overlapsize:segment_size - overlapsize]), -1)
            out[:, pointer:] += y[..., :nl]
            return out
