"""Seeded synthetic parameters for benchmarks and serving warm-up: the reference's shapes and initialisation
scale (reference model.py:114-120,163-167) with perturbed BatchNorm affine / running statistics so that eval-mode
normalisation is exercised.  Draws the same numpy RandomState stream as the test oracle's generator (a CPU test
pins the two together), so bench.py measures the very parameters the bench-size golden fixture was made with --
without importing anything from oracle/."""
from __future__ import annotations

import math
from typing import Dict

import numpy as np

from .engine import STAGE_CHANNELS


def synthetic_state_dict(seed: int, num_classes: int = 1211, n_stages: int = 4) -> Dict[str, np.ndarray]:
    rs = np.random.RandomState(seed)
    sd: Dict[str, np.ndarray] = {}
    cin = 1
    for s in range(n_stages):
        c, i = STAGE_CHANNELS[s], s + 1
        for name, co, ci, k in ((f"model.conv{i}", c, cin, 5), (f"model.layer{i}.0.conv1", c, c, 3),
                                (f"model.layer{i}.0.conv2", c, c, 3)):
            sd[name + ".weight"] = (rs.randn(co, ci, k, k) * math.sqrt(2.0 / (k * k * co))).astype(np.float32)
            bn = name.replace("conv", "bn")
            sd[bn + ".weight"] = rs.uniform(0.5, 1.5, co).astype(np.float32)
            sd[bn + ".bias"] = (rs.randn(co) * 0.1).astype(np.float32)
            sd[bn + ".running_mean"] = (rs.randn(co) * 0.1).astype(np.float32)
            sd[bn + ".running_var"] = rs.uniform(0.5, 1.5, co).astype(np.float32)
            sd[bn + ".num_batches_tracked"] = np.zeros((), np.int64)
        cin = c
    for name, n_out, n_in in (("model.fc", 512, 2048), ("model.classifier", num_classes, 512)):
        k = 1.0 / math.sqrt(float(n_in))
        sd[name + ".weight"] = rs.uniform(-k, k, (n_out, n_in)).astype(np.float32)
        sd[name + ".bias"] = rs.uniform(-k, k, n_out).astype(np.float32)
    return sd
