#!/usr/bin/env python3
"""Per-layer timing and tile plans of the eval forward at small batch (the near-tie refinement forward and the
serving case).  python tools/small_batch_probe.py [B ...]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from deepspeaker_pytorch_amd._native import ConvShape
from deepspeaker_pytorch_amd.model import DeepSpeakerModel, get_engine
from deepspeaker_pytorch_amd.synthetic import synthetic_state_dict

dev = torch.device("cuda", 0)
sd = synthetic_state_dict(0, 8)
eng = get_engine()
LAYERS = [(80, 32, 64, 64, 3, 1), (80, 32, 64, 128, 5, 2), (40, 16, 128, 128, 3, 1), (40, 16, 128, 256, 5, 2),
          (20, 8, 256, 256, 3, 1), (20, 8, 256, 512, 5, 2), (10, 4, 512, 512, 3, 1)]
for B in [int(a) for a in sys.argv[1:]] or [1, 24]:
    x = torch.randn(B, 1, 160, 64, device=dev)
    for prec in ("bf16x3", "f16"):
        m = DeepSpeakerModel(512, 8, precision=prec)
        m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
        m = m.to(dev).eval()
        with torch.no_grad():
            for _ in range(5):
                m(x)
            eng.profile = []
            for _ in range(10):
                m(x)
            torch.cuda.synchronize()
            by = {}
            for label, fl, e0, e1, _ in eng.profile:
                d = by.setdefault(label, [0.0, 0])
                d[0] += e0.elapsed_time(e1)
                d[1] += 1
            eng.profile = None
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(50):
                m(x)
            t1.record()
            torch.cuda.synchronize()
        print(f"== B={B} {prec}: forward {t0.elapsed_time(t1) / 50 * 1e3:.0f} us")
        for (h, w, ci, co, k, s_) in LAYERS:
            out8 = (ctypes.c_int * 8)()
            shp = ConvShape(B, h, w, ci, co, k, s_)
            if prec == "f16":
                eng.lib.call("ds_conv_f16_plan_describe", ctypes.byref(shp), out8)
            else:
                eng.lib.call("ds_conv_bf16_plan_describe", ctypes.byref(shp), 1, out8)
            ho, wo = (h - 1) // s_ + 1, (w - 1) // s_ + 1
            label = f"conv{k}x{k}s{s_}_{ci}to{co}_{ho}x{wo}"
            t = by[label]
            print(f"   {label:30s} {t[0] / t[1] * 1e3:7.1f} us   tile {out8[0]}x{out8[1]} rt {out8[2]} ni {out8[3]} wgs {out8[4]} "
                  f"lds {out8[5]} thr {out8[6]} x {out8[7]}")
