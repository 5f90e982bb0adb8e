#!/usr/bin/env python3
"""The split-operand bf16 convolution planner's tile configuration against every other one the layer fits
(ds_conv_bf16_set_forced_cfg), train-mode form (raw output + BatchNorm partial sums), at one member's 256 utterances and at
the whole step's 768; rounds interleaved in one process, outputs compared bitwise.     python tools/bf16_cfg_ab.py"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch

from deepspeaker_pytorch_amd._native import ConvShape
from deepspeaker_pytorch_amd.model import get_engine
from conv_probe import LAYERS

eng = get_engine()
dev = torch.device("cuda:0")
setcfg = eng.lib.raw("ds_conv_bf16_set_forced_cfg")
NAMES = ["128x64 4w", "160x128 4w", "256x64 4w", "160x128 2w", "160x256 4w", "320x128 4w", "320x64 2w", "128x128 2w", "128x256 4w"]
rounds = 6
for B in ([int(a) for a in sys.argv[1:]] or [256, 768]):
    for name, H, W, Cin, Cout, KS, s in LAYERS:
        x = torch.randn(B, H, W, Cin, device=dev)
        w = torch.randn(Cout, Cin, KS, KS, device=dev) * 0.05
        bank = eng._pack_bf16(w, KS)
        Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
        fl = 2.0 * B * Ho * Wo * Cout * Cin * KS * KS
        shp = ConvShape(B, H, W, Cin, Cout, KS, s)
        variants = []
        for cfg in [-1] + list(range(9)):
            setcfg(cfg)
            out8 = (ctypes.c_int * 8)()
            if eng.lib.raw("ds_conv_bf16_plan_describe")(ctypes.byref(shp), 1, out8) != 0:
                continue
            try:
                y, st = eng.conv_bf16(x, bank, True, B, H, W, Cin, Cout, KS, s, want_stats=True)
            except Exception:
                continue
            variants.append((cfg, list(out8), [], y))
        torch.cuda.synchronize()
        for rd in range(rounds):
            for cfg, d, ts, _ in variants:
                setcfg(cfg)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(3):
                    eng.conv_bf16(x, bank, True, B, H, W, Cin, Cout, KS, s, want_stats=True)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3 / 3)
        setcfg(-1)
        print(f"B = {B}  {name}")
        for cfg, d, ts, y in variants:
            t = np.array(ts[1:])
            same = torch.equal(y, variants[0][3])
            print(f"   {'planner' if cfg < 0 else 'cfg %d %s' % (cfg, NAMES[cfg]):22s} {np.median(t):8.1f} us  {fl / np.median(t) * 1e-6:5.0f} TF  tile {d[0]}x{d[1]} thr {d[6]} RT {d[2]} NI {d[3]} "
                  f"wgs {d[4]}  {'=' if same else 'DIFFERS'}")
