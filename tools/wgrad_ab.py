#!/usr/bin/env python3
"""Same-process A/B of the split-operand bf16 filter-gradient kernel between library builds (tools/f16_ab.py builds
them): per training-step layer at the 768-utterance size, interleaved rounds, median us.
    python tools/f16_ab.py --build-only base: old:-DDS_WGRAD_GSL3=4,-DDS_WGRAD_XSL3=12
    gpurun -- python tools/wgrad_ab.py base old"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch

from deepspeaker_pytorch_amd._native import ConvShape, NativeLib
from conv_probe import LAYERS

names = sys.argv[1:] or ["base"]
libs = [NativeLib(os.path.join(ROOT, "tools", "_ab", f"libds_ab_{n}.so")) for n in names]
dev = torch.device("cuda:0")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: ctypes.c_void_p(t.data_ptr())
B = 768
print("layer".ljust(18) + "".join(n.rjust(26) for n in names))
for name, H, W, Cin, Cout, KS, s in LAYERS:
    if Cin % 64:
        continue
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    x = torch.randn(B, H, W, Cin, device=dev)
    gy = torch.randn(B, Ho, Wo, Cout, device=dev)
    shp = ConvShape(B, H, W, Cin, Cout, KS, s)
    fl = 2.0 * B * Ho * Wo * Cout * Cin * KS * KS
    outs, ts = [], [[] for _ in libs]
    wss = [torch.empty(lib.raw("ds_conv_wgrad_bf16_workspace_floats")(ctypes.byref(shp)), device=dev) for lib in libs]
    gws = [torch.empty(Cout, Cin, KS, KS, device=dev) for _ in libs]
    for rnd in range(7):
        for i, lib in enumerate(libs):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3):
                lib.call("ds_conv_wgrad_bf16", ctypes.byref(shp), p(x), p(gy), p(wss[i]), p(gws[i]), st)
            e1.record()
            torch.cuda.synchronize()
            if rnd:
                ts[i].append(e0.elapsed_time(e1) * 1e3 / 3)
    row = name.ljust(18)
    for i in range(len(libs)):
        err = float((gws[i] - gws[0]).norm() / gws[0].norm())
        row += f"{np.median(ts[i]):9.1f} us {fl / np.median(ts[i]) * 1e-6:6.0f} TF {err:.0e}".rjust(26)
    print(row)
