#!/usr/bin/env python3
"""What the near-tie refinement's f32-class forward costs by itself: DeepSpeakerModel.embed_reference (split-operand bf16)
on 12 .. 768 rows, alone on the chip, events around 10 forwards.     python tools/refine_cost_probe.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from deepspeaker_pytorch_amd.model import DeepSpeakerModel, get_engine
from deepspeaker_pytorch_amd.synthetic import synthetic_state_dict

dev = torch.device("cuda", 0)
m = DeepSpeakerModel(512, 16, precision="f16")
m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in synthetic_state_dict(0, 16).items()})
m = m.to(dev).eval()
eng = get_engine()
for rows in (12, 24, 48, 96, 192, 384, 768):
    x = torch.randn(rows, 1, 160, 64, device=dev)
    with torch.no_grad():
        for _ in range(3):
            m.embed_reference(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            m.embed_reference(x)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        eng.profile = []
        m.embed_reference(x)
        torch.cuda.synchronize()
        prof, eng.profile = eng.profile, None
        conv = sum(p[2].elapsed_time(p[3]) for p in prof)
    if rows in (12, 96, 768):
        for label, fl, a, b, _ in prof:
            t = a.elapsed_time(b) * 1e3
            print(f"        {label:34s} {t:8.1f} us  {fl / t / 1e6:6.0f} TF (x3 on the matrix cores)")
    print(f"{rows:4d} rows: {ms * 1e3:8.1f} us per forward = {ms * 1e3 / rows:6.2f} us per row ({len(prof)} convolution launches, {conv * 1e3:.0f} us "
          f"inside them with event pairs); as a share of a 768-row fp16 forward of 1750 us: {ms * 1e3 / 1750:.1%}")
