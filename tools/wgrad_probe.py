"""Filter-gradient kernels per ResCNN layer at B=256 (one of the three forwards of a training step):
f32 matrix cores vs split-operand bf16.  python tools/wgrad_probe.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from deepspeaker_pytorch_amd import _native  # noqa: E402
from conv_probe import LAYERS  # noqa: E402

lib = _native.load()
dev = torch.device("cuda:0")
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: ctypes.c_void_p(t.data_ptr())
B = 256
for name, H, W, Cin, Cout, KS, s in LAYERS:
    Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
    x = torch.randn(B, H, W, Cin, device=dev)
    gy = torch.randn(B, Ho, Wo, Cout, device=dev)
    gw = torch.empty(Cout, Cin, KS, KS, device=dev)
    shp = _native.ConvShape(B, H, W, Cin, Cout, KS, s)
    fl = 2.0 * B * Ho * Wo * Cout * Cin * KS * KS
    out = []
    for fn, wsfn, extra in (("ds_conv_wgrad_f32", "ds_conv_wgrad_workspace_floats", (0,)),
                            ("ds_conv_wgrad_bf16", "ds_conv_wgrad_bf16_workspace_floats", ())):
        ws = torch.empty(lib.raw(wsfn)(ctypes.byref(shp)), device=dev)
        run = lambda: lib.call(fn, ctypes.byref(shp), p(x), p(gy), p(ws), p(gw), *extra, st)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        out.append(f"{us:8.1f}us {fl / us * 1e-6:6.1f}TF")
    print(f"{name:16s} f32 {out[0]}   bf16x3 {out[1]}")
