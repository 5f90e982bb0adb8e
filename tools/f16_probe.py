#!/usr/bin/env python3
"""GPU probe of the fp16 path at the bench size: accuracy against the exact-f32 path (embeddings, distances,
filter decisions) and per-layer timings of both low-precision paths.  python tools/f16_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from deepspeaker_pytorch_amd.model import DeepSpeakerModel, get_engine
from deepspeaker_pytorch_amd.synthetic import synthetic_state_dict

dev = torch.device("cuda", 0)
sd_np = synthetic_state_dict(0, 1211)
g = torch.Generator(device="cpu").manual_seed(1234)
x = torch.randn(768, 1, 160, 64, generator=g).to(dev)
embs = {}
for prec in ("f32", "bf16x3", "f16"):
    m = DeepSpeakerModel(512, 1211, precision=prec)
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd_np.items()})
    m = m.to(dev).eval()
    with torch.no_grad():
        e = m(x).clone()
        torch.cuda.synchronize()
        eng = get_engine()
        for _ in range(3):
            m(x)
        eng.profile = []
        for _ in range(5):
            m(x)
        torch.cuda.synchronize()
        by = {}
        for label, fl, e0, e1, _ in eng.profile:
            d = by.setdefault(label, [0.0, 0.0])
            d[0] += fl
            d[1] += e0.elapsed_time(e1)
        eng.profile = None
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(10):
            m(x)
        t1.record()
        torch.cuda.synchronize()
    embs[prec] = e.double().cpu()
    tot_f, tot_t = sum(v[0] for v in by.values()), sum(v[1] for v in by.values())
    print(f"== {prec}: forward {t0.elapsed_time(t1) / 10:.3f} ms / 768 utt; conv family {tot_t / 5:.3f} ms, "
          f"{tot_f / tot_t / 1e9:.1f} TFLOP/s algorithmic")
    for k, v in by.items():
        print(f"   {k:32s} {v[1] / 5 * 1e3 / (2 if '3x3' in k else 1):8.1f} us/launch  {v[0] / v[1] / 1e9:7.1f} TF")
ref = embs["f32"]


def dist(a, b):
    return torch.sqrt(((a - b) ** 2).sum(1) + 1e-4 / 512)


gap_ref = dist(ref[:256], ref[512:]) - dist(ref[:256], ref[256:512]) - 0.1
print("f32: min |gap| %.3e, selected %d" % (gap_ref.abs().min(), int((gap_ref < 0).sum())))
for prec in ("bf16x3", "f16"):
    e = embs[prec]
    rel = ((e - ref).norm(dim=1) / ref.norm(dim=1))
    gap = dist(e[:256], e[512:]) - dist(e[:256], e[256:512]) - 0.1
    print(f"{prec}: rel-L2 mean {rel.mean():.3e} max {rel.max():.3e}; max|d|/max {((e - ref).abs().max() / ref.abs().max()):.3e}; "
          f"gap err max {(gap - gap_ref).abs().max():.3e}; flips {int(((gap < 0) != (gap_ref < 0)).sum())}")
