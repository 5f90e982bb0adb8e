#!/usr/bin/env python3
"""The fused BasicBlock kernel (ds_conv_block_f16) against the same block as two single-layer launches of the persistent
fp16 convolution, stages 1 and 2 at the bench size, rounds interleaved in one process, outputs compared bitwise.
    python tools/f16_block_vs_layers.py [--rounds 12]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from deepspeaker_pytorch_amd._native import ConvShape, DS_EPI_AFFINE, DS_EPI_CLIP, DS_EPI_RESIDUAL
from deepspeaker_pytorch_amd.model import get_engine

rounds = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 12
eng = get_engine()
dev = torch.device("cuda", 0)
B = 768
st = eng._stream(torch.zeros(1, device=dev))
setcfg = eng.lib.raw("ds_conv_f16_set_forced_cfg")
for (h, w, c) in [(80, 32, 64), (40, 16, 128)]:
    x = torch.randn(B, h, w, c, device=dev).abs().half()
    w1, w2 = (torch.randn(c, c, 3, 3, device=dev) * (1.0 / (c * 9) ** 0.5) for _ in range(2))
    p1, p2 = eng._pack_f16(w1, 3), eng._pack_f16(w2, 3)
    s1, h1, s2, h2 = (torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev), torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev))
    shp = ConvShape(B, h, w, c, c, 3, 1)
    mid = torch.empty(B, h, w, c, dtype=torch.float16, device=dev)
    y_two = torch.empty_like(mid)
    y_blk = torch.empty_like(mid)

    def two(cfg=-1):
        setcfg(cfg)
        eng.lib.call("ds_conv_fwd_f16", ctypes.byref(shp), eng._p(x), eng._p(p1), eng._p(s1), eng._p(h1), None, eng._p(mid),
                     DS_EPI_AFFINE | DS_EPI_CLIP, st)
        eng.lib.call("ds_conv_fwd_f16", ctypes.byref(shp), eng._p(mid), eng._p(p2), eng._p(s2), eng._p(h2), eng._p(x), eng._p(y_two),
                     DS_EPI_AFFINE | DS_EPI_CLIP | DS_EPI_RESIDUAL, st)
        setcfg(-1)

    def blk():
        eng.lib.call("ds_conv_block_f16", eng._p(x), eng._p(p1), eng._p(p2), eng._p(s1), eng._p(h1), eng._p(s2), eng._p(h2),
                     eng._p(y_blk), B, h, w, c, 0, st)
    variants = [("fused block", blk)]
    for cfg in [-1] + list(range(7)):
        setcfg(cfg)
        out8 = (ctypes.c_int * 8)()
        rc = eng.lib.raw("ds_conv_f16_plan_describe")(ctypes.byref(shp), out8)
        setcfg(-1)
        if rc == 0 and out8[7] >= 10000:
            variants.append((f"two launches, {'planner' if cfg < 0 else 'cfg %d' % cfg} ({out8[0]}x{out8[1]}, {out8[6]} thr)", lambda cfg=cfg: two(cfg)))
    times = [[] for _ in variants]
    for _, fn in variants:
        for _ in range(3):
            fn()
    torch.cuda.synchronize()
    blk()
    two()
    torch.cuda.synchronize()
    assert torch.equal(y_two, y_blk)
    for rd in range(rounds):
        for vi, (_, fn) in enumerate(variants):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                fn()
            e1.record()
            torch.cuda.synchronize()
            times[vi].append(e0.elapsed_time(e1) * 250.0)
    fl = 2 * 2.0 * B * h * w * c * c * 9
    print(f"BasicBlock {c} channels on {h}x{w} maps (bitwise equal)")
    for (name, _), t in zip(variants, times):
        t = np.array(t)
        print(f"   {name:48s} {np.median(t):8.1f} us [{t.min():7.1f}]  {fl / np.median(t) / 1e6:6.0f} TF")
