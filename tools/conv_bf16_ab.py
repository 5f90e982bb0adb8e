#!/usr/bin/env python3
"""Same-process A/B of the split-operand bf16 forward convolution (train-mode form: raw output + statistics) between
library builds (tools/f16_ab.py --build-only builds them), per 3x3 / 5x5 layer at one member's 256 utterances.
    gpurun -- python tools/conv_bf16_ab.py [B] base other ..."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch

from deepspeaker_pytorch_amd._native import ConvShape, NativeLib
from deepspeaker_pytorch_amd.engine import Engine
from conv_probe import LAYERS

B = 256
args = sys.argv[1:]
if args and args[0].isdigit():
    B = int(args.pop(0))
names = args or ["base"]
engs = [Engine(NativeLib(os.path.join(ROOT, "tools", "_ab", f"libds_ab_{n}.so"))) for n in names]
dev = torch.device("cuda:0")
print("layer".ljust(18) + "".join(n.rjust(40) for n in names))
for name, H, W, Cin, Cout, KS, s in LAYERS:
    x = torch.randn(B, H, W, Cin, device=dev)
    w = torch.randn(Cout, Cin, KS, KS, device=dev) * 0.05
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    fl = 2.0 * B * Ho * Wo * Cout * Cin * KS * KS
    banks = [e._pack_bf16(w, KS) for e in engs]
    ts, outs = [[] for _ in engs], []
    for rnd in range(7):
        for i, e in enumerate(engs):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                y, st = e.conv_bf16(x, banks[i], True, B, H, W, Cin, Cout, KS, s, want_stats=True)
            e1.record()
            torch.cuda.synchronize()
            if rnd:
                ts[i].append(e0.elapsed_time(e1) * 1e3 / 3)
            if rnd == 6:
                outs.append(y)
    row = name.ljust(18)
    for i, e in enumerate(engs):
        out8 = (ctypes.c_int * 8)()
        shp = ConvShape(B, H, W, Cin, Cout, KS, s)
        e.lib.call("ds_conv_bf16_plan_describe", ctypes.byref(shp), 1, out8)
        same = torch.equal(outs[i], outs[0])
        row += f"{np.median(ts[i]):8.1f} us {fl / np.median(ts[i]) * 1e-6:5.0f} TF tile {out8[0]}x{out8[1]} thr {out8[6]} {'=' if same else '!'}".rjust(40)
    print(row)
