"""Deterministic weight / input recipe for the real-width parity fixtures (fixture plan 3).

TEST INFRASTRUCTURE ONLY.  A cfg32 state_dict is 36-95 MB, too big to commit, so the fixture stores only outputs and the
recipe below regenerates identical weights on both sides (the reference in oracle/make_golden.py, the HIP modules and the
CPU oracle in tests/): each state_dict key gets its own CPU generator seeded from crc32(key), so the fill does not depend
on module registration order or on either side's own initialisation code.
"""
import math
import zlib

import torch


def _gen(key: str, seed: int) -> torch.Generator:
    return torch.Generator().manual_seed((zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)


def fill_state(state: dict, seed: int) -> dict:
    """New state_dict with the same keys/shapes, seeded values, non-trivial everywhere (no zero-init gains)."""
    out = {}
    for key, v in state.items():
        g = _gen(key, seed)
        leaf = key.split(".")[-1]
        if leaf == "out_gain":
            t = torch.full(v.shape, 0.7)
        elif leaf == "alpha_txt":
            t = torch.full(v.shape, 0.3)
        elif leaf in ("rel_pos_bias", "pos_emb"):
            t = 0.3 * torch.randn(v.shape, generator=g)
        elif leaf == "freqs":
            t = 2 * math.pi * torch.randn(v.shape, generator=g)
        elif leaf == "phases":
            t = 2 * math.pi * torch.rand(v.shape, generator=g)
        elif leaf == "weights":                                   # MP_Conv: magnitude is normalised away
            t = torch.randn(v.shape, generator=g)
        elif leaf == "weight" and v.ndim == 1:                    # GroupNorm / LayerNorm affine
            t = 1.0 + 0.2 * torch.randn(v.shape, generator=g)
        elif leaf == "bias":
            t = 0.2 * torch.randn(v.shape, generator=g)
        elif leaf == "weight":                                    # nn.Conv2d patch embedding
            t = torch.randn(v.shape, generator=g) / math.sqrt(v[0].numel())
        else:
            raise KeyError(f"recipe has no rule for state_dict key {key!r}")
        out[key] = t.to(v.dtype)
    return out


def make_inputs(B: int, C: int, R: int, E: int, text_len: int, text_dim: int, seed: int) -> dict:
    """Synthetic step inputs in the shape of reference Utils/training.py:125-153 (log-spaced sigmas, partial masks)."""
    g = _gen("inputs", seed)
    x0 = 0.5 * torch.randn(B, C, R, R, generator=g)
    sigma = torch.logspace(math.log10(0.02), math.log10(40.0), B).view(B, 1, 1, 1)
    x = x0 + sigma * torch.randn(B, C, R, R, generator=g)
    text = torch.randn(B, text_len, text_dim, generator=g)
    um = (torch.rand(B, E, generator=g) > 0.3).float()
    vm = (torch.rand(B, E, generator=g) > 0.3).float()
    um[:, 0] = 1.0                                               # at least top_k=2 experts stay eligible per sample
    um[:, E - 1] = 1.0
    vm[:, 1] = 1.0
    vm[:, E - 2] = 1.0
    return dict(x0=x0, sigma=sigma, x=x, text=text, unet_mask=um, vit_mask=vm)
