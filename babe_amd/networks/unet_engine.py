"""CQTDiff+ UNet body on the babe_hip kernels: forward and hand-wired input-VJP.

Mirrors the wiring of /root/reference/networks/cqtdiff+.py:746-839 (forward) and
ResnetBlock.forward :452-493; the VJP replaces torch.autograd through the UNet
(/root/reference/testing/blind_bwe_sampler.py:120).  No torch compute ops are used on
the data path: torch only allocates buffers; every tensor op is a C-ABI call (ops.py).

Layout: activations are [B, C, F, T] fp32, T contiguous.  Octave/skip concatenations are
never materialised as copies of big tensors: producers write straight into frequency
sub-views of the consumer's buffer and channel concatenation is a two-source conv input.
Each dilation layer keeps exactly one tensor (its input) for the VJP; GroupNorm / FiLM /
GELU are recomputed in the backward kernels.
"""
import math
import os

import torch

from .. import ops

RS2 = 1.0 / math.sqrt(2.0)
# the library-side sequencer (csrc/unet_engine.hip): one C call per direction.  BABE_UNET_C=1 enables it (fp32 networks).
USE_C = os.environ.get("BABE_UNET_C", "0") == "1"
MERGE_TAIL = os.environ.get("BABE_MERGE_TAIL", "1") != "0"      # 0: N -> N block VJP stores gz and merges with axpby2 (A/B switch)


class _Block:
    """One ResnetBlock: packed weights + per-call saved tensors."""

    def __init__(self, sd, prefix, num_dils, film_index, proj_after=False, precision="f32"):
        self.p = prefix
        self.nd = num_dils
        self.proj_after = proj_after
        g = lambda k: sd.get(prefix + k)
        PC = lambda w: ops.PackedConv(w, precision)
        self.proj_in = PC(g("proj_in.weight")) if g("proj_in.weight") is not None else None
        self.res_conv = PC(g("res_conv.weight")) if g("res_conv.weight") is not None else None
        self.proj_out = PC(g("proj_out.weight")) if (proj_after and g("proj_out.weight") is not None) else None
        self.H = [PC(g(f"H.{d}.weight")) for d in range(num_dils)]
        self.gamma = [g(f"norm.{d}.gamma").reshape(-1).contiguous() for d in range(num_dils)]
        self.N = self.H[0].Cout
        self.k53 = self.H[0].KH > 1
        self.film_off = []            # (affine offset, gate offset) into the batched FiLM output
        for d in range(num_dils):
            self.film_off.append((film_index.add(g(f"affine.{d}.weight"), g(f"affine.{d}.bias")),
                                  film_index.add(g(f"gate.{d}.weight"), g(f"gate.{d}.bias"))))
        self.saved = None

    def dil(self, d):
        return 2 ** d if self.k53 else 1


class _FilmIndex:
    """Collects every FiLM Linear (affine/gate of every dilation layer) into one [J,emb] matrix."""

    def __init__(self):
        self.W, self.b, self.J = [], [], 0

    def add(self, W, b):
        off = self.J
        self.W.append(W)
        self.b.append(b)
        self.J += W.shape[0]
        return off

    def finalize(self):
        self.Wcat = torch.cat(self.W, 0).contiguous()
        self.bcat = torch.cat(self.b, 0).contiguous()
        self.W = self.b = None


class UnetEngine:
    def __init__(self, sd, Ns, num_dils, num_octs=7, bins_per_oct=64, precision="f32"):
        """sd: dict of DEVICE fp32 tensors with the reference's state_dict key names.
        precision: conv arithmetic, 'f32' (exact fp32 MFMA, the parity path), 'bf16x3' or 'bf16'."""
        self.precision = precision
        self.Ns, self.num_dils, self.nocts, self.bpo = list(Ns), list(num_dils), num_octs, bins_per_oct
        self.dev = sd["embedding.RFF_freq"].device
        fi = _FilmIndex()
        self.emb_W = [(sd[f"embedding.MLP.{i}.weight"].contiguous(), sd[f"embedding.MLP.{i}.bias"].contiguous()) for i in range(3)]
        self.rff_freq = sd["embedding.RFF_freq"].reshape(-1).contiguous()
        self.init_blk, self.main_blk, self.pyr_conv = [], [], []
        for i in range(num_octs):
            self.init_blk.append(_Block(sd, f"downs.{i}.0.", 1, fi, precision=precision))
            self.pyr_conv.append(ops.PackedConv(sd[f"downs.{i}.1.weight"], precision))
            self.main_blk.append(_Block(sd, f"downs.{i}.2.", num_dils[i], fi, precision=precision))
        self.mid_blk = _Block(sd, "middle.0.1.", num_dils[-1], fi, precision=precision)
        self.mid_out = _Block(sd, "middle.0.0.", 1, fi, proj_after=True, precision=precision)
        self.up_out, self.up_blk = [], []
        for i in range(num_octs):
            j = num_octs - 1 - i
            self.up_out.append(_Block(sd, f"ups.{i}.0.", 1, fi, proj_after=True, precision=precision))
            self.up_blk.append(_Block(sd, f"ups.{i}.1.", num_dils[j], fi, precision=precision))
        fi.finalize()
        self.film_idx = fi
        self._scratch = {}

    def clone_state(self):
        """A second engine over the SAME packed weights with its own per-call state (saved activations, scratch), so that
        two batch items can run on two streams at once."""
        import copy
        c = copy.copy(self)
        c._scratch = {}
        c.__dict__.pop("_cunet", None)                   # own state + workspace over the SAME plan
        if self.__dict__.get("_cunet") is not None:
            c._cunet_parent = self._cunet
        elif USE_C and self.precision == "f32":
            c._cunet_parent = self._c_engine()
        cp = lambda blks: [copy.copy(b) for b in blks]
        c.init_blk, c.main_blk, c.up_out, c.up_blk = cp(self.init_blk), cp(self.main_blk), cp(self.up_out), cp(self.up_blk)
        c.mid_blk, c.mid_out = copy.copy(self.mid_blk), copy.copy(self.mid_out)
        return c

    # ------------------------------------------------------------------ helpers
    def buf(self, *shape):
        return torch.empty(*shape, device=self.dev, dtype=torch.float32)

    def scratch(self, name, numel):
        t = self._scratch.get(name)
        if t is None or t.numel() < numel:
            t = torch.empty(numel, device=self.dev, dtype=torch.float32)
            self._scratch[name] = t
        return t[:numel]

    def scratch_i16(self, name, numel):
        t = self._scratch.get(name)
        if t is None or t.numel() < numel:
            t = torch.empty(numel, device=self.dev, dtype=torch.int16)
            self._scratch[name] = t
        return t[:numel]

    def embed(self, cnoise):
        """cnoise [B,1] -> FiLM vectors for every layer [B, J] (RFF_MLP_Block + all affine/gate Linears)."""
        h = ops.rff(cnoise, self.rff_freq)
        for W, b in self.emb_W:
            h = ops.linear(h, W, b, relu=True)
        return ops.linear(h, self.film_idx.Wcat, self.film_idx.bcat, relu=False)

    def _film(self, film, off, N):
        return film[:, off:off + N]

    # ------------------------------------------------------------------ ResnetBlock
    def block_fwd(self, blk, x, film, out, x2=None):
        """out <- ResnetBlock(cat(x,x2)); out may be a strided frequency sub-view."""
        B, _, Fq, T = x.shape
        N = blk.N
        if blk.proj_in is not None:
            z = ops.conv2d(x, blk.proj_in, self.buf(B, N, Fq, T), x2=x2)
        else:
            assert x2 is None
            z = x if x.is_contiguous() else ops.axpby(x, self.buf(B, N, Fq, T))
        saved = []
        # precision='bf16': the GELU output goes to the conv as bf16 units (half the bytes, both conv operands by LDS-DMA)
        units = blk.nd > 0 and ops.units_ok(blk.H[0], N, N, T) and z.is_contiguous()
        if units:
            au = self.scratch_i16("au", B * ops.lib().babe_units_size(N, Fq, T) * 8)
        zsum = None                                          # (part, S): GroupNorm sums of z formed by the conv that wrote it
        for d in range(blk.nd):
            aoff, goff = blk.film_off[d]
            gate = self._film(film, goff, N).contiguous()
            znew = self.buf(B, N, Fq, T)
            ua = ops.units_args(au, blk.H[d], znew, N, dil=blk.dil(d), res=z, oscale=gate, alpha=RS2, rbeta=RS2) if units else None
            if ua is not None and ops.units_supported(ua):       # the library's verdict, not a Python guess
                stats, scale = ops.gn_scale(z, blk.gamma[d], self._film(film, aoff, N))
                ops.scale_gelu_units(z, scale, au)
                ops.conv2d_units(au, blk.H[d], znew, N, args=ua)
                zsum = None
            else:
                a = self.scratch("a", B * N * Fq * T).view(B, N, Fq, T)
                stats, scale = ops.gn_scale_gelu(z, blk.gamma[d], self._film(film, aoff, N), a, fused=zsum)  # (finalize inside the GELU launch)
                # the next layer's GroupNorm reads znew: its sums come out of this conv's epilogue when the F(4,5) kernel runs it
                zsum = ops.conv2d(a, blk.H[d], znew, dil=blk.dil(d), res=z, oscale=gate, alpha=RS2, rbeta=RS2,
                                  fwd_stat=(N // 8) if d + 1 < blk.nd else None)
                if d + 1 >= blk.nd:
                    zsum = None
            saved.append((z, stats, scale, gate))
            z = znew
        if blk.proj_out is not None:
            z = ops.conv2d(z, blk.proj_out, self.buf(B, blk.proj_out.Cout, Fq, T))
        if blk.res_conv is not None:
            ops.conv2d(x, blk.res_conv, out, x2=x2, res=z, alpha=RS2, rbeta=RS2)
        else:
            ops.axpby2(z, x, out, RS2, RS2)                  # (x + h)/sqrt2 in one pass
        blk.saved = saved
        return out

    def block_vjp(self, blk, g_out, g_in, accumulate=False, consume=False):
        """g_in (+)= VJP of the block w.r.t. its (concatenated) input. g_out: [B,Cout,F,T] (may be strided).
        consume=True: g_out is a dense buffer owned by the caller that may be overwritten (saves a full copy)."""
        B, _, Fq, T = g_out.shape
        N = blk.N
        beta = 1.0 if accumulate else 0.0
        if (blk.res_conv is None and blk.proj_out is None and blk.proj_in is None and not accumulate and blk.nd > 0
                and g_out.is_contiguous() and ops.AXPBY2):
            # N -> N block (every main block of the encoder, the middle block): the residual path's RS2*g_out and the main path's
            # c*gz are merged in ONE pass at the end (12 instead of 8 + 12 bytes per element).  For that g_out has to survive the
            # chain: the first layer's gn_bwd reads it as its residual input and writes into a buffer of its own (same traffic),
            # the later layers update that buffer in place.  Same arithmetic, same rounding as the two-pass form.
            gz = self.buf(B, N, Fq, T) if blk.nd > 1 else None
            da = self.scratch("a", B * N * Fq * T).view(B, N, Fq, T)
            src = g_out
            merged = g_in.is_contiguous() and MERGE_TAIL
            for d in reversed(range(blk.nd)):
                z, stats, scale, gate = blk.saved[d]
                fs = ops.conv2d(src, blk.H[d], da, dil=blk.dil(d), transpose=True, in_scale=gate, alpha=RS2, vjp_stat=(z, scale, N // 8))
                if d == 0 and merged:
                    # the last layer's VJP pass writes g_in = RS2*g_out + RS2*gz itself (gz is never stored)
                    ops.gn_bwd(z, da, src, scale, stats, g_in, RS2, merge=(g_out, RS2, RS2), fused=fs)
                else:
                    if gz is None:
                        gz = self.buf(B, N, Fq, T)
                    ops.gn_bwd(z, da, src, scale, stats, gz, RS2, fused=fs)
                    src = gz
            if not merged:
                ops.axpby2(g_out, gz, g_in, RS2, RS2)
            blk.saved = None
            return g_in
        # residual path
        if blk.res_conv is not None:
            ops.conv2d(g_out, blk.res_conv, g_in, transpose=True, alpha=RS2, rbeta=beta, res=g_in if accumulate else None)
        else:
            ops.axpby(g_out, g_in, alpha=RS2, beta=beta)
        # main path: gradient w.r.t. z_last.  The chain below is linear in gz, so the 1/sqrt2 of the block's
        # output merge is carried as a scalar `c` and applied once at the end instead of scaling a copy.
        c = 1.0
        if blk.proj_out is not None:
            gz = ops.conv2d(g_out, blk.proj_out, self.buf(B, N, Fq, T), transpose=True, alpha=RS2)
        elif consume and g_out.is_contiguous():
            gz, c = g_out, RS2
        else:
            gz = ops.axpby(g_out, self.buf(B, N, Fq, T), alpha=RS2)
        da = self.scratch("a", B * N * Fq * T).view(B, N, Fq, T)
        for d in reversed(range(blk.nd)):
            z, stats, scale, gate = blk.saved[d]
            fs = ops.conv2d(gz, blk.H[d], da, dil=blk.dil(d), transpose=True, in_scale=gate, alpha=RS2, vjp_stat=(z, scale, N // 8))
            ops.gn_bwd(z, da, gz, scale, stats, gz, RS2, fused=fs)
        if blk.proj_in is not None:
            ops.conv2d(gz, blk.proj_in, g_in, transpose=True, res=g_in, alpha=c, rbeta=1.0)
        else:
            ops.axpby(gz, g_in, alpha=c, beta=1.0)
        blk.saved = None
        return g_in

    # ------------------------------------------------------------------ forward
    def _c_engine(self):
        """The library-side sequencer for this engine STATE (networks/unet_c.py), or None: fp32 only, BABE_UNET_C=0 switches it
        off, the measurement hook's per-launch events work with either."""
        if not USE_C or self.precision != "f32":
            return None
        cu = self.__dict__.get("_cunet")
        if cu is None:
            from .unet_c import CUnet
            parent = self.__dict__.get("_cunet_parent")
            cu = parent.clone(self) if parent is not None else CUnet(self)
            self._cunet = cu
        return cu

    def forward(self, C_list, film):
        """C_list[j]: planar [B,2,bpo,T_j], index 0 = lowest octave. Returns same structure."""
        cu = self._c_engine()
        if cu is not None:
            self._c_fwd = True
            return cu.fwd([c.contiguous() for c in C_list], film)
        self._c_fwd = False
        n, bpo, Ns = self.nocts, self.bpo, self.Ns
        B = C_list[0].shape[0]
        Ts = [C_list[n - 1 - i].shape[-1] for i in range(n)]      # level i time length
        self.Ts, self.B = Ts, B
        hs, pyrs = [], []
        XC = self.buf(B, Ns[0], bpo, Ts[0])
        for i in range(n):
            C = C_list[n - 1 - i]
            Fi = bpo * (i + 1)
            self.block_fwd(self.init_blk[i], C, film, XC[:, :, :bpo, :])
            # pyramid side path
            if i == 0:
                pyr = ops.resample(C, self.buf(B, 2, bpo, Ts[0] // 2), 0)
            elif i < n - 1:
                pyr_new = self.buf(B, 2, Fi, Ts[i] // 2)
                ops.resample(C, pyr_new[:, :, :bpo, :], 0)
                ops.resample(pyrs[-1], pyr_new[:, :, bpo:, :], 0)
                pyr = pyr_new
            else:
                pyr_new = self.buf(B, 2, Fi, Ts[i])
                ops.axpby(C, pyr_new[:, :, :bpo, :])
                ops.axpby(pyrs[-1], pyr_new[:, :, bpo:, :])
                pyr = pyr_new
            pyrs.append(pyr)
            H = self.block_fwd(self.main_blk[i], XC, film, self.buf(B, Ns[i], Fi, Ts[i]))
            hs.append(H)
            if i < n - 1:
                XCn = self.buf(B, Ns[i], Fi + bpo, Ts[i + 1])
                sub = XCn[:, :, bpo:, :]
                ops.resample(H, sub, 0)
                ops.conv2d(pyr, self.pyr_conv[i], sub, res=sub, alpha=RS2, rbeta=RS2)
                XC = XCn
            else:
                X = ops.conv2d(pyr, self.pyr_conv[i], self.buf(B, Ns[i], Fi, Ts[i]), res=H, alpha=RS2, rbeta=RS2)
        self.hs = hs
        X = self.block_fwd(self.mid_blk, X, film, self.buf(*X.shape))
        Xout = self.block_fwd(self.mid_out, X, film, self.buf(B, 2, bpo * n, Ts[-1]))
        outs = [None] * n
        for i in range(n):
            j = n - 1 - i
            Fj = bpo * (j + 1)
            Nout = Ns[max(j - 1, 0)]
            R = self.block_fwd(self.up_blk[i], X, film, self.buf(B, Nout, Fj, Ts[j]), x2=hs[j])
            O = self.block_fwd(self.up_out[i], R, film, self.buf(B, 2, Fj, Ts[j]))
            ops.axpby(O, Xout, alpha=RS2, beta=RS2)               # Xout <- (Xout + O)/sqrt2
            outs[i] = ops.axpby(Xout[:, :, :bpo, :], self.buf(B, 2, bpo, Ts[j]))
            if j > 0:
                X = ops.resample(R[:, :, bpo:, :], self.buf(B, Nout, Fj - bpo, Ts[j - 1]), 1)
                Xout = ops.resample(Xout[:, :, bpo:, :], self.buf(B, 2, Fj - bpo, Ts[j - 1]), 1)
        return outs

    # ------------------------------------------------------------------ input-VJP
    def vjp(self, gouts):
        """gouts[i]: gradient w.r.t. outs[i] (index 0 = lowest octave). Returns gradients w.r.t. C_list."""
        if getattr(self, "_c_fwd", False):
            return self._c_engine().vjp([g.contiguous() for g in gouts])
        n, bpo, Ns, Ts, B = self.nocts, self.bpo, self.Ns, self.Ts, self.B
        gH = [None] * n
        gX_prev = gXO_prev = None          # gradients w.r.t. X_{j-1}, XO_{j-1} (outputs of the up-samplers)
        for j in range(n):                  # reverse of the decoder order (which ran j = n-1 .. 0)
            i = n - 1 - j
            Fj = bpo * (j + 1)
            Nout = Ns[max(j - 1, 0)]
            gXOp = self.buf(B, 2, Fj, Ts[j])
            ops.axpby(gouts[i], gXOp[:, :, :bpo, :])
            gR = self.buf(B, Nout, Fj, Ts[j])
            if j > 0:
                ops.resample(gXO_prev, gXOp[:, :, bpo:, :], 3)
                gR[:, :, :bpo, :].zero_()
                ops.resample(gX_prev, gR[:, :, bpo:, :], 3)
                accumulate = True
            else:
                accumulate = False
            # O_j = up_out(R_j) entered Xout as rs2*O_j
            gO = ops.axpby(gXOp, self.buf(B, 2, Fj, Ts[j]), alpha=RS2)
            self.block_vjp(self.up_out[i], gO, gR, accumulate=accumulate, consume=True)
            gXO_prev = ops.axpby(gXOp, self.buf(B, 2, Fj, Ts[j]), alpha=RS2)
            gcat = self.buf(B, 2 * Ns[j], Fj, Ts[j])
            self.block_vjp(self.up_blk[i], gR, gcat, consume=True)
            gX_prev = gcat[:, :Ns[j]]
            gH[j] = gcat[:, Ns[j]:]
        # middle: Xout_6 = mid_out(M); X_6 = M
        gM = self.buf(B, Ns[-1], bpo * n, Ts[-1])
        ops.axpby(gX_prev, gM)
        self.block_vjp(self.mid_out, gXO_prev, gM, accumulate=True, consume=True)
        gXm = self.block_vjp(self.mid_blk, gM, self.buf(*gM.shape), consume=True)
        # encoder
        gC = [None] * n
        gpyr_next = None                   # gradient flowing into pyr_i from level i+1
        gP = None
        for i in reversed(range(n)):
            Fi = bpo * (i + 1)
            if i == n - 1:
                gHi = ops.axpby2(gH[i], gXm, self.buf(B, Ns[i], Fi, Ts[i]), 1.0, RS2)
                gpyr = ops.conv2d(gXm, self.pyr_conv[i], self.buf(B, 2, Fi, Ts[i]), transpose=True, alpha=RS2)
            else:
                # P_i = (down(H_i) + pconv_i(pyr_i)) * rs2 lives in XC_{i+1}[:, :, bpo:, :]
                gHi = ops.resample(gP, self.buf(B, Ns[i], Fi, Ts[i]), 2, alpha=RS2, beta=1.0, res=gH[i])   # g_skip + rs2 down^T(g_P), one pass
                gpyr = self.buf(B, 2, Fi, Ts[i] // 2)
                ops.conv2d(gP, self.pyr_conv[i], gpyr, transpose=True, alpha=RS2,
                           res=gpyr_next if gpyr_next is not None else None, rbeta=1.0 if gpyr_next is not None else 0.0)
            Nin = Ns[max(i - 1, 0)]
            gXC = self.block_vjp(self.main_blk[i], gHi, self.buf(B, Nin, Fi, Ts[i]), consume=True)
            gCi = self.buf(B, 2, bpo, Ts[i])
            self.block_vjp(self.init_blk[i], gXC[:, :, :bpo, :], gCi)
            gP = gXC[:, :, bpo:, :] if i > 0 else None
            # pyramid
            if i == n - 1:
                ops.axpby(gpyr[:, :, :bpo, :], gCi, beta=1.0)
                gpyr_next = gpyr[:, :, bpo:, :]
            elif i > 0:
                ops.resample(gpyr[:, :, :bpo, :], gCi, 2, beta=1.0)
                gpyr_next = ops.resample(gpyr[:, :, bpo:, :], self.buf(B, 2, Fi - bpo, Ts[i]), 2)
            else:
                ops.resample(gpyr, gCi, 2, beta=1.0)
            gC[n - 1 - i] = gCi
        self.hs = None
        return gC
