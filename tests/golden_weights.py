"""Seeded weights shared by tests/golden/make_golden.py (which imports the reference) and the parity tests (which must
not): only OUTPUTS are stored in the fixtures, the weights are re-derived here from seeds."""
import torch

FULL_NS = [64, 96, 96, 128, 128, 256, 256]
FULL_DILS = [2, 3, 4, 5, 6, 7, 7]


def scale_gates(sd, seed=5):
    """init_zero gates (1e-7) make residual branches numerically invisible: re-randomise gates, GroupNorm gammas and
    FiLM biases to O(1) (same rule as make_golden.scale_gates)."""
    g = torch.Generator().manual_seed(seed)
    for k in sd:
        if ".gate." in k:
            sd[k] = torch.randn(sd[k].shape, generator=g) * (0.1 if k.endswith("weight") else 0.5)
        if ".norm." in k and k.endswith("gamma"):
            sd[k] = 1.0 + 0.2 * torch.randn(sd[k].shape, generator=g)
        if ".affine." in k and k.endswith("bias"):
            sd[k] = 0.2 * torch.randn(sd[k].shape, generator=g)
    return sd


def full_width_sd(seed=0):
    """Weights of the full-width goldens (unet_full_*.npz): babe_amd's init_state_dict(seed) - the benchmark's weights,
    reference init rule with O(1) gates - with gates / norm gammas / FiLM biases re-randomised by scale_gates."""
    from babe_amd.networks.cqtdiff_plus import init_state_dict
    return scale_gates(init_state_dict(FULL_NS, FULL_DILS, seed=seed, gate_scale=1.0), seed=5)
