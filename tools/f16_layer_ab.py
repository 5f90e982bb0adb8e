#!/usr/bin/env python3
"""Per-layer A/B of the fp16 convolution's kernel choices at the bench size (768 utterances), in one process, rounds
interleaved: persistent workgroups vs one tile per workgroup (DS_CONV_HINT_NO_PERSIST), 32- vs 16-channel chunks
(DS_CONV_HINT_CHUNK16).  Outputs of the variants of one layer are compared bitwise (chunk width and persistence do not
change the arithmetic).     python tools/f16_layer_ab.py [--rounds 15]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from deepspeaker_pytorch_amd._native import (ConvShape, DS_CONV_HINT_CHUNK16, DS_CONV_HINT_NO_PERSIST, DS_CONV_HINT_NO_WIDE, DS_CONV_HINT_ONE_QUEUE, DS_CONV_IN_PLANES16,
                                             DS_EPI_AFFINE, DS_EPI_CLIP, DS_EPI_RESIDUAL)
from deepspeaker_pytorch_amd.model import get_engine

rounds = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 15
eng = get_engine()
dev = torch.device("cuda", 0)
B = 768
# (H, W, Cin, Cout, KS, stride, residual, plane-major input)
LAYERS = [(80, 32, 64, 128, 5, 2, False, True), (40, 16, 128, 256, 5, 2, False, False), (20, 8, 256, 256, 3, 1, True, False),
          (20, 8, 256, 512, 5, 2, False, False), (10, 4, 512, 512, 3, 1, True, False)]
VARIANTS = [("persistent", 0), ("persistent, one tile queue", DS_CONV_HINT_ONE_QUEUE),
            ("persistent, 64-wide register tiles", DS_CONV_HINT_NO_WIDE),
            ("persistent, 64-wide, one tile queue", DS_CONV_HINT_NO_WIDE | DS_CONV_HINT_ONE_QUEUE), ("one tile / wg", DS_CONV_HINT_NO_PERSIST), ("persistent, 16-ch chunks", DS_CONV_HINT_CHUNK16),
            ("one tile / wg, 16-ch chunks", DS_CONV_HINT_CHUNK16 | DS_CONV_HINT_NO_PERSIST)]
st = eng._stream(torch.zeros(1, device=dev))
for (h, w, ci, co, k, s_, res, planes) in LAYERS:
    ho, wo = (h - 1) // s_ + 1, (w - 1) // s_ + 1
    x = torch.randn(B, h, w, ci, device=dev).abs().half()
    xin = x
    if planes:      # [Cin/16][B*H*W][16]
        xin = x.view(B * h * w, ci // 16, 16).permute(1, 0, 2).contiguous()
    wt = torch.randn(co, ci, k, k, device=dev) * (1.0 / (ci * k * k) ** 0.5)
    wp = eng._pack_f16(wt, k)
    sc, sh = torch.rand(co, device=dev) + 0.5, torch.randn(co, device=dev)
    r = (torch.randn(B, ho, wo, co, device=dev).abs() * 4).half() if res else None
    shp = ConvShape(B, h, w, ci, co, k, s_)
    base = DS_EPI_AFFINE | DS_EPI_CLIP | (DS_EPI_RESIDUAL if res else 0) | (DS_CONV_IN_PLANES16 if planes else 0)
    variants = [v for v in VARIANTS if k == 5 or not (v[1] & DS_CONV_HINT_CHUNK16)]
    if not (k == 3 and co % 256 == 0):
        variants = [v for v in variants if not (v[1] & DS_CONV_HINT_NO_WIDE)]       # (the hint changes nothing there)
    if "--only" in sys.argv and f"{ci}x{co}x{k}" not in sys.argv[sys.argv.index("--only") + 1].split(","):
        continue
    if planes:
        variants = [v for v in variants if v[1] & DS_CONV_HINT_CHUNK16]        # plane-major input implies 16-channel chunks
    if k == 5:
        variants += [(n + ", one tile queue", hnt | DS_CONV_HINT_ONE_QUEUE) for n, hnt in variants if not (hnt & DS_CONV_HINT_NO_PERSIST)]
    outs, times, descs = [], [[] for _ in variants], []
    for name, hint in variants:
        out8 = (ctypes.c_int * 8)()
        eng.lib.call("ds_conv_f16_plan_describe_hinted", ctypes.byref(shp), base | hint, out8)
        descs.append(list(out8))
        y = torch.empty(B, ho, wo, co, dtype=torch.float16, device=dev)
        for _ in range(3):
            eng.lib.call("ds_conv_fwd_f16", ctypes.byref(shp), eng._p(xin), eng._p(wp), eng._p(sc), eng._p(sh), eng._p(r), eng._p(y),
                         base | hint, st)
        outs.append(y)
    torch.cuda.synchronize()
    for rd in range(rounds):
        for vi, (name, hint) in enumerate(variants):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            eng.lib.call("ds_conv_fwd_f16", ctypes.byref(shp), eng._p(xin), eng._p(wp), eng._p(sc), eng._p(sh), eng._p(r),
                         eng._p(outs[vi]), base | hint, st)
            e1.record()
            torch.cuda.synchronize()
            times[vi].append(e0.elapsed_time(e1) * 1e3)
    fl = 2.0 * B * ho * wo * co * ci * k * k
    print(f"conv{k}x{k}s{s_} {ci}->{co} {ho}x{wo}" + (" (plane-major input)" if planes else ""))
    for vi, (name, hint) in enumerate(variants):
        t = np.array(times[vi])
        same = torch.equal(outs[vi], outs[0])
        d = descs[vi]
        print(f"   {name:42s} {np.median(t):8.1f} us [{t.min():7.1f}]  {fl / np.median(t) / 1e6:6.0f} TF   tile {d[0]}x{d[1]} RT {d[2]} NI {d[3]} "
              f"tiles {d[4]} plan {d[7]}   {'bitwise equal' if same else 'DIFFERS'}")
        assert same
