"""Shared helpers for golden fixtures (TEST INFRASTRUCTURE, see oracle/tatt_oracle.py header).

``randomize_state_dict`` perturbs a freshly initialised reference-format state_dict so
that parity tests are well conditioned: the reference's default init leaves BatchNorm /
LayerNorm affine at (1, 0), every bias of the STN head at 0 and ``stn_fc2.weight`` at 0
(model/stn_head.py:59-90), which would hide whole sub-graphs from a parity check.  The
function is deterministic (CPU generator) and is applied identically to the reference
module (tools/gen_golden.py) and to the oracle / HIP path (tests).
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch

BUFFER_KEYS = ("infoGen.pe.pe", "tps.inverse_kernel", "tps.padding_matrix",
               "tps.target_coordinate_repr", "tps.target_control_points")


def randomize_state_dict(sd: Dict[str, torch.Tensor], seed: int = 7) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k in sorted(sd.keys()):
        v = sd[k]
        if k in BUFFER_KEYS or k.endswith("num_batches_tracked"):
            out[k] = v.clone()
            continue
        r = torch.randn(v.shape, generator=g, dtype=torch.float32)
        if k.endswith("running_var"):
            out[k] = 1.0 + 0.5 * torch.rand(v.shape, generator=g)
        elif k.endswith("running_mean"):
            out[k] = 0.1 * r
        elif k == "stn_head.stn_fc2.weight":
            out[k] = 0.02 * r
        elif k == "stn_head.stn_fc2.bias":
            out[k] = v + 0.02 * r
        elif v.numel() > 1 and bool((v == 1).all()):
            out[k] = 1.0 + 0.1 * r
        elif v.numel() > 1 and bool((v == 0).all()):
            out[k] = 0.05 * r
        else:
            out[k] = v.clone()
    return {k: out[k] for k in sd.keys()}


def summarize(t: torch.Tensor) -> np.ndarray:
    """(l2 norm, sum, first 4, last 4) of a tensor as float64 -- a compact fingerprint used
    where the full tensor would be too large to commit (e.g. the 4.7 M-element GRU grads)."""
    f = t.detach().double().reshape(-1)
    head = torch.zeros(4, dtype=torch.float64)
    tail = torch.zeros(4, dtype=torch.float64)
    n = min(4, f.numel())
    head[:n] = f[:n]
    tail[:n] = f[-n:]
    return torch.cat([torch.stack([f.norm(), f.sum()]), head, tail]).numpy()


def make_inputs(B: int, H: int = 16, W: int = 64, seed: int = 0, scale: int = 2):
    """Synthetic inputs of SURVEY.md §8d: x U[0,1), text prior softmax(randn), hr U[0,1)."""
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, 4, H, W, generator=g)
    tp = torch.softmax(torch.randn(B, 37, 1, 26, generator=g), 1)
    hr = torch.rand(B, 4, H * scale, W * scale, generator=g)
    return x, tp, hr
