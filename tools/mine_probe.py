"""Timing of the semi-hard negative search (256 local anchors) against the all-gathered candidate set of
1 / 2 / 4 / 8 ranks (768 embeddings each) -- the part of a bench.py step whose cost grows with --gpus."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from deepspeaker_pytorch_amd.mining import mine_semihard_negatives  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)
for world in (1, 2, 4, 8):
    n, m, d = 256, 768 * world, 512
    a = torch.nn.functional.normalize(torch.randn(n, d, generator=g), dim=1).mul(10).to(dev)
    p = torch.nn.functional.normalize(torch.randn(n, d, generator=g), dim=1).mul(10).to(dev)
    c = torch.nn.functional.normalize(torch.randn(m, d, generator=g), dim=1).mul(10).to(dev)
    la = torch.randint(0, 64, (n,), generator=g).to(dev)
    lc = torch.randint(0, 64, (m,), generator=g).to(dev)
    for _ in range(3):
        mine_semihard_negatives(a, p, la, c, lc)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        idx, dist = mine_semihard_negatives(a, p, la, c, lc)
    e1.record()
    torch.cuda.synchronize()
    print(f"world {world}: {n} anchors x {m} candidates: {e0.elapsed_time(e1) * 1e3 / 20:8.1f} us per search "
          f"(incl. d_p and allocations), checksum {int(idx.sum())}")
