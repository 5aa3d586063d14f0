"""Long-recording driver: fixed-length segments + Hann cross-fade, as the reference's harness does it.

Restates the non-AR path of ``BlindTester.formal_test_bwe`` (/root/reference/testing/blind_bwe_tester.py:
421-566): segments of ``exp.audio_len`` samples starting every ``segL - discard_end - OLA`` samples
(discard_end = 200, ``tester.formal_test.OLA`` = 256), the last one zero-padded; each restored segment keeps
its first ``segL - discard_end`` samples, is faded with the two halves of a 2*OLA Hann window and
overlap-added.  The reference restores the segments one at a time; they are independent, so here they are
batched (per-clip semantics) and can be sharded over GPUs.  Host-side plumbing only (slicing / cross-fade);
the sampler does the work.
"""
import torch


def plan_segments(L, segL, discard_end=200, ola=256):
    """[(start, n_valid)] : start sample of each segment and how many of its samples exist in the file."""
    hop = segL - discard_end - ola
    starts = [0]
    ix = hop
    while ix < L - segL - discard_end:          # :469 (discard_start = 0)
        starts.append(ix)
        ix += hop
    if L > segL - discard_end - ola or len(starts) == 1:
        starts.append(ix)                        # the final, possibly short, segment (:521-566)
    if starts[-1] >= L:                          # file shorter than one hop: a single (padded) segment
        starts = [0]
    return [(s, min(segL, L - s)) for s in starts]


def cut_segments(y, segL, plan):
    """y [L] -> [S, segL] (zero-padded tail)."""
    segs = torch.zeros(len(plan), segL, device=y.device, dtype=y.dtype)
    for i, (s, n) in enumerate(plan):
        segs[i, :n] = y[s:s + n]
    return segs


def assemble(preds, plan, L, segL, discard_end=200, ola=256):
    """Cross-fade the restored segments [S, segL] back into a file of L samples."""
    out = torch.zeros(L, device=preds.device, dtype=preds.dtype)
    w = torch.hann_window(2 * ola, device=preds.device, dtype=preds.dtype)
    S = len(plan)
    for i, (s, n) in enumerate(plan):
        last = i == S - 1
        keep = n if last else segL - discard_end
        if S == 1:
            keep = min(n, L)
        p = preds[i, :keep].clone()
        if i > 0:
            p[:ola] *= w[:ola]
        if not last:
            p[-ola:] *= w[ola:]
        end = min(s + keep, L)
        out[s:end] += p[: end - s]
    return out


def restore_file(sampler, y, batch_size=8, blind=True, filt=None):
    """y [L] device tensor -> (restored [L], [(start, end, filter_params)]) using sampler.predict_blind_bwe
    (or predict_bwe(filt, 'fc_A') when blind=False) on batches of segments."""
    segL = sampler.args.exp.audio_len
    ola = sampler.args.tester.get("formal_test", {}).get("OLA", 256) if hasattr(sampler.args.tester, "get") else 256
    L = y.shape[-1]
    plan = plan_segments(L, segL, 200, ola)
    segs = cut_segments(y, segL, plan)
    preds, filters = [], []
    for i in range(0, segs.shape[0], batch_size):
        chunk = segs[i:i + batch_size].contiguous()
        if blind:
            x, fp = sampler.predict_blind_bwe(chunk)
            fp = fp if fp.dim() == 3 else fp.unsqueeze(0).expand(chunk.shape[0], -1, -1)
            filters += [fp[j] for j in range(chunk.shape[0])]
        else:
            x = sampler.predict_bwe(chunk, filt, "fc_A")
        preds.append(x)
    preds = torch.cat(preds, 0)
    out = assemble(preds, plan, L, segL, 200, ola)
    return out, [((s, s + segL), f) for (s, _), f in zip(plan, filters)]


def restore_file_AR(sampler, y, filt, filt_type="fc_A", overlap_s=0.25, discard_end=200):
    """Autoregressive out-painting of a whole recording with a KNOWN (previously estimated) filter, as the second
    half of BlindTester.test_real_blind_bwe_complete does (/root/reference/testing/blind_bwe_tester.py:786-859):
    the first segment is restored with predict_bwe, every following segment starts `overlap` samples inside the
    previous result, which is passed as already-known signal (mask = 1 there) to predict_bwe_AR.  Sequential by
    construction; y [L] device tensor -> restored [L]."""
    segL = sampler.args.exp.audio_len
    overlap = int(overlap_s * sampler.args.exp.sample_rate)
    L = y.shape[-1]
    dev = y.device
    out = torch.zeros(L, device=dev)
    if L <= segL:
        seg = torch.zeros(1, segL, device=dev)
        seg[0, :L] = y
        return sampler.predict_bwe(seg, filt, filt_type)[0, :L]
    ix = 0
    pred = sampler.predict_bwe(y[ix:ix + segL].unsqueeze(0).contiguous(), filt, filt_type)
    prev = pred[..., : segL - discard_end]
    out[ix:ix + segL - discard_end] = prev[0]
    ix += segL - overlap - discard_end
    y_masked = torch.zeros(1, segL, device=dev)
    mask = torch.ones(1, segL, device=dev)
    mask[..., overlap:] = 0
    while ix < L - segL - discard_end:
        y_masked[..., :overlap] = prev[..., segL - overlap - discard_end:]
        pred = sampler.predict_bwe_AR(y[ix:ix + segL].unsqueeze(0).contiguous(), y_masked, filt, filt_type, mask=mask)
        prev = pred[..., : segL - discard_end]
        out[ix:ix + segL - discard_end] = prev[0]
        ix += segL - overlap - discard_end
    seg = y[ix:]
    n = seg.shape[-1]
    y_masked[..., :overlap] = pred[..., -overlap:]           # (the reference takes the tail of the last prediction, :842)
    if n < segL:
        seg_zp = torch.zeros(1, segL, device=dev)
        seg_zp[0, :n] = seg
        y_masked[..., n:] = 0
        mask[..., n:] = 0
    else:
        seg_zp = seg[:segL].unsqueeze(0).contiguous()
        n = segL
    pred = sampler.predict_bwe_AR(seg_zp, y_masked, filt, filt_type, mask=mask)
    out[ix:ix + n] = pred[0, :n]
    return out


def rate_plan(fs, sample_rate, denoiser_rate=None):
    """The sample-rate conversions of the file-level flows as a list of (from, to) pairs, in order, with the string
    "denoise" where the denoiser runs - ONE rule for `restore_recording_complete` and `python -m babe_amd.restore`:
      * no denoiser: fs -> exp.sample_rate (testing/blind_bwe_tester.py:410; nothing at equal rates);
      * with a denoiser, the reference's own rule (testing/denoise_and_bwe_tester.py:279-289): fs -> denoiser rate before it
        and denoiser rate -> exp.sample_rate after it, BOTH only `if fs != sample_rate_denoiser`.  The quirk that comes with
        it is kept, not repaired: a file already AT the denoiser's rate reaches the model unconverted even when the model's
        rate differs - exactly what the reference does with such a file."""
    if denoiser_rate is None:
        return [(fs, sample_rate)] if fs != sample_rate else []
    if fs == denoiser_rate:
        return ["denoise"]
    return [(fs, denoiser_rate), "denoise", (denoiser_rate, sample_rate)]


def restore_recording_complete(sampler, degraded, *, n_segments_blindstep=1, ix_start=0, std=0.1, overlap_s=0.25,
                               typefilter="fc_A", denoiser=None, rng=None, fs=None, sample_rate_denoiser=None):
    """Whole-recording flow of BlindTester.test_real_blind_bwe_complete (/root/reference/testing/blind_bwe_tester.py:
    710-867; with the denoiser pre-pass: testing/denoise_and_bwe_tester.py:248-411, config #5):

      1. optional denoiser pre-pass over the whole file (`denoiser.apply_denoiser`, :279-285 - sequential, not joint);
      2. normalise the file to `complete_recording.std` (:296-297);
      3. BLIND step (:307-329): estimate the low-pass filter on `n_segments_blindstep` segments - the one at
         `ix_start` seconds if 1, else that many segments at `rng.randint(0, L - segL)` (numpy's global generator in the
         reference) - in ONE batch with the reference's batch coupling (one filter fitted on the flattened batch,
         whole-batch guidance norm), whatever `sampler.batch_semantics` is set to for independent clips;
      4. NON-BLIND autoregressive pass over the file with that filter (:350-407, `restore_file_AR`);
      5. undo the normalisation (:409).

    degraded: [L] or [1, L] device tensor sampled at `fs` (None: already at exp.sample_rate, no conversion).  With `fs` given
    the reference's rate conversions are part of the flow (babe_amd/resample.py = torchaudio.functional.resample as published),
    exactly where testing/denoise_and_bwe_tester.py:279-289 has them: with a denoiser, fs -> sample_rate_denoiser before it and
    sample_rate_denoiser -> exp.sample_rate after it - BOTH only `if fs != sample_rate_denoiser`, as the reference writes it (a
    file already at the denoiser's rate goes to the model unconverted there too); without one, fs -> exp.sample_rate.
    Returns (restored [L'] at exp.sample_rate, estimated_filter [2, K], blind-step prediction [n, segL])."""
    import numpy as np
    rng = rng or np.random
    args = sampler.args
    segL = args.exp.audio_len
    d = degraded.reshape(1, -1).float()
    srd = None
    if denoiser is not None:
        srd = sample_rate_denoiser if sample_rate_denoiser is not None else denoiser._get("sample_rate_denoiser")
    # (fs None: the caller states that the file is already where it has to be - only the denoiser runs)
    for step in (rate_plan(fs, args.exp.sample_rate, srd) if fs is not None else (["denoise"] if denoiser is not None else [])):
        if step == "denoise":
            d = denoiser.apply_denoiser(d)
        else:
            from ..resample import resample
            d = resample(d, step[0], step[1])
    s = d.std(-1)
    d = std * d / s.unsqueeze(-1)
    L = d.shape[-1]
    ix_first = int(args.exp.sample_rate * ix_start)
    if n_segments_blindstep == 1:
        y = d[..., ix_first:ix_first + segL]
    else:
        y = d[..., ix_first:ix_first + segL].repeat(n_segments_blindstep, 1)
        for j in range(n_segments_blindstep):
            ix = int(rng.randint(0, L - segL))
            y[j] = d[0, ix:ix + segL]
    keep = sampler.batch_semantics
    sampler.batch_semantics = "reference"
    try:
        pred, estimated_filter = sampler.predict_blind_bwe(y.contiguous())
    finally:
        sampler.batch_semantics = keep
    out = restore_file_AR(sampler, d[0], estimated_filter, typefilter, overlap_s=overlap_s)
    return out * (s / std), estimated_filter, pred
