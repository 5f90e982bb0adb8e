#!/usr/bin/env python3
"""Is the fp16 convolution planner's tile configuration the fastest one?  For every single-layer launch of the forward at
the bench size (768 utterances) each of the seven configurations the layer fits is forced in turn
(ds_conv_f16_set_forced_cfg), rounds interleaved in one process, outputs compared bitwise with the planner's own choice.
    python tools/f16_cfg_ab.py [--rounds 12]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from deepspeaker_pytorch_amd._native import ConvShape, DS_CONV_IN_PLANES16, DS_EPI_AFFINE, DS_EPI_CLIP, DS_EPI_RESIDUAL
from deepspeaker_pytorch_amd.model import get_engine

rounds = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 12
eng = get_engine()
dev = torch.device("cuda", 0)
B = 768
NAMES = ["160x128 2w", "160x256 4w", "320x128 4w", "320x64 2w", "128x128 2w", "128x256 4w", "640x64 4w"]
# (H, W, Cin, Cout, KS, stride, residual, plane-major input)
LAYERS = [(80, 32, 64, 128, 5, 2, False, True), (40, 16, 128, 256, 5, 2, False, False), (20, 8, 256, 256, 3, 1, True, False),
          (20, 8, 256, 512, 5, 2, False, False), (10, 4, 512, 512, 3, 1, True, False)]
if "--train-shapes" in sys.argv:      # the single-layer launches only the fp16 TRAINING step has: the 3x3 layers of stages 1-2 (their
    # eval forward is the fused block) and the 5x5 stride-2 data gradients run as 3x3 convolutions with 4 Cin output channels
    LAYERS = [(80, 32, 64, 64, 3, 1, False, False), (40, 16, 128, 128, 3, 1, False, False), (40, 16, 128, 256, 3, 1, False, False),
              (20, 8, 256, 512, 3, 1, False, False), (10, 4, 512, 1024, 3, 1, False, False)]
st = eng._stream(torch.zeros(1, device=dev))
setcfg = eng.lib.raw("ds_conv_f16_set_forced_cfg")
for (h, w, ci, co, k, s_, res, planes) in LAYERS:
    ho, wo = (h - 1) // s_ + 1, (w - 1) // s_ + 1
    x = torch.randn(B, h, w, ci, device=dev).abs().half()
    xin = x.view(B * h * w, ci // 16, 16).permute(1, 0, 2).contiguous() if planes else x
    wt = torch.randn(co, ci, k, k, device=dev) * (1.0 / (ci * k * k) ** 0.5)
    wp = eng._pack_f16(wt, k)
    sc, sh = torch.rand(co, device=dev) + 0.5, torch.randn(co, device=dev)
    r = (torch.randn(B, ho, wo, co, device=dev).abs() * 4).half() if res else None
    shp = ConvShape(B, h, w, ci, co, k, s_)
    base = DS_EPI_AFFINE | DS_EPI_CLIP | (DS_EPI_RESIDUAL if res else 0) | (DS_CONV_IN_PLANES16 if planes else 0)
    variants = []
    for cfg in [-1] + list(range(7)):
        setcfg(cfg)
        out8 = (ctypes.c_int * 8)()
        rc = eng.lib.raw("ds_conv_f16_plan_describe_hinted")(ctypes.byref(shp), base, out8)
        if rc != 0:
            continue
        y = torch.empty(B, ho, wo, co, dtype=torch.float16, device=dev)
        rc = eng.lib.raw("ds_conv_fwd_f16")(ctypes.byref(shp), eng._p(xin), eng._p(wp), eng._p(sc), eng._p(sh), eng._p(r), eng._p(y), base, st)
        if rc != 0:
            continue
        for _ in range(2):
            eng.lib.raw("ds_conv_fwd_f16")(ctypes.byref(shp), eng._p(xin), eng._p(wp), eng._p(sc), eng._p(sh), eng._p(r), eng._p(y), base, st)
        variants.append((cfg, list(out8), y, []))
    torch.cuda.synchronize()
    for rd in range(rounds):
        for cfg, d, y, ts in variants:
            setcfg(cfg)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                eng.lib.raw("ds_conv_fwd_f16")(ctypes.byref(shp), eng._p(xin), eng._p(wp), eng._p(sc), eng._p(sh), eng._p(r), eng._p(y), base, st)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 250.0)
    setcfg(-1)
    fl = 2.0 * B * ho * wo * co * ci * k * k
    print(f"conv{k}x{k}s{s_} {ci}->{co} {ho}x{wo}" + (" (plane-major input)" if planes else ""))
    for cfg, d, y, ts in variants:
        t = np.array(ts)
        same = torch.equal(y, variants[0][2])
        print(f"   {'planner' if cfg < 0 else 'cfg %d %s' % (cfg, NAMES[cfg]):22s} {np.median(t):8.1f} us [{t.min():7.1f}]  {fl / np.median(t) / 1e6:6.0f} TF   "
              f"tile {d[0]}x{d[1]} RT {d[2]} NI {d[3]} tiles {d[4]} thr {d[6]} plan {d[7]}   {'bitwise equal' if same else 'DIFFERS'}")
        assert same
