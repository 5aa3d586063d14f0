"""Checkpoint and wav I/O conventions of the reference ("next" row 4).

load_checkpoint: the three layouts BlindTester.load_checkpoint accepts
(/root/reference/testing/blind_bwe_tester.py:238-272): state['ema'] (a state_dict), or state['ema_weights']
(a list) zipped against state['model'] keys, or the same list zipped against the trainable entries only.
write_audio_file: /root/reference/utils/logging.py:297-320 (mono, float) - soundfile is not available in this
image, scipy.io.wavfile writes the same 32-bit float PCM.  Filter estimates are pickled per file like
blind_bwe_tester.py:466,576.
"""
import os
import pickle

import numpy as np
import torch


def ema_state_dict(state, reference_keys=None):
    """Extract the EMA weights from a reference checkpoint dict as {key: tensor}."""
    if "ema" in state and isinstance(state["ema"], dict):
        return dict(state["ema"])
    if "ema_model" in state:
        return dict(state["ema_model"])
    if "ema_weights" in state and "model" in state:
        keys = list(state["model"].keys())
        ema = list(state["ema_weights"])
        if len(ema) == len(keys):
            return {k: t for k, t in zip(keys, ema)}
        out, i = {}, 0                      # list holds the trainable tensors only (third fall-back, :259-269)
        for k, t in state["model"].items():
            if getattr(t, "requires_grad", False) or (reference_keys is not None and k in reference_keys and not k.endswith(".kernel") and k != "embedding.RFF_freq"):
                out[k] = ema[i]
                i += 1
            else:
                out[k] = t
        return out
    if "model" in state:
        return dict(state["model"])
    return dict(state)                      # a bare state_dict


def load_checkpoint(network, path, map_location="cpu"):
    """network.load_state_dict(EMA weights of `path`); returns the iteration counter (0 if absent)."""
    state = torch.load(path, map_location=map_location, weights_only=False)
    sd = ema_state_dict(state, set(network.state_dict().keys()))
    network.load_state_dict(sd)
    return int(state.get("it", 0)) if isinstance(state, dict) else 0


def read_audio_file(path):
    """-> (float32 tensor [L] (mono mix-down), sample_rate)."""
    from scipy.io import wavfile
    sr, x = wavfile.read(path)
    x = np.asarray(x)
    if x.dtype.kind == "i":
        x = x.astype(np.float32) / float(np.iinfo(x.dtype).max + 1)
    elif x.dtype.kind == "u":
        x = (x.astype(np.float32) - 128.0) / 128.0
    x = x.astype(np.float32)
    if x.ndim == 2:
        x = x.mean(axis=1)
    return torch.from_numpy(x), int(sr)


def write_audio_file(x, sr, string, path="tmp"):
    """Mono float wav `path/string.wav`; returns the file path (same contract as utils/logging.write_audio_file)."""
    from scipy.io import wavfile
    os.makedirs(path, exist_ok=True)
    p = os.path.join(path, string + ".wav")
    wavfile.write(p, int(sr), x.detach().flatten().cpu().numpy().astype(np.float32))
    return p


def write_filter_data(filter_data, path, name):
    """[((start, end), filter_params)] -> path/name.filter_data.pkl (blind_bwe_tester.py:466,576)."""
    os.makedirs(path, exist_ok=True)
    p = os.path.join(path, name + ".filter_data.pkl")
    with open(p, "wb") as f:
        pickle.dump([((int(a), int(b)), fp.detach().cpu()) for (a, b), fp in filter_data], f)
    return p
