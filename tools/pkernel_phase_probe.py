#!/usr/bin/env python3
"""Where a workgroup of the PERSISTENT fp16 convolution (conv_mfma_f16_pkernel) spends a tile, at HEAD: a probe copy of the
library (-DDS_F16_PROBE: s_memtime stamps at the tile top / after the staging barrier / after the last chunk / after the
epilogue's barrier / at the epilogue's end, first four tiles of every workgroup) runs the five single-layer launches of the
forward at B = 768.  Prints per-phase medians in s_memtime ticks and as fractions of the tile, the spread over workgroups of
the time a workgroup finishes, and the launch duration.
    build container: python tools/pkernel_phase_probe.py --build-only ;  GPU box: python tools/pkernel_phase_probe.py"""
import ctypes
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "deepspeaker-pytorch_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "_ab", "libds_pkprobe.so")

if "--build-only" in sys.argv:
    objd = os.path.join(ROOT, "tools", "_ab", "obj_pkprobe")
    os.makedirs(objd, exist_ok=True)
    srcs = [s for s in sorted(glob.glob(os.path.join(CSRC, "*.hip"))) if "conv_mfma_f16" in s or "bn_pack" in s]

    def cc(s):
        o = os.path.join(objd, os.path.basename(s)[:-4] + ".o")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DDS_F16_PROBE", "-mllvm",
                        "-pragma-unroll-threshold=1000000", "-Wno-pass-failed", f"-I{CSRC}", f"-I{ROOT}/include", "-c", "-o", o, s], check=True)
        return o
    with ThreadPoolExecutor(16) as ex:
        objs = list(ex.map(cc, srcs))
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT, *objs], check=True)
    print(OUT)
    sys.exit(0)

import numpy as np
import torch

from deepspeaker_pytorch_amd._native import ConvShape, DS_CONV_IN_PLANES16, DS_EPI_AFFINE, DS_EPI_CLIP, DS_EPI_RESIDUAL

dll = ctypes.CDLL(OUT)
# the probe copy of the library draws its tile counters from a zeroed buffer of ours, like the product wrapper does
dll.ds_sched_workspace_bytes.restype = ctypes.c_size_t
dll.ds_sched_set_workspace.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
_sched_ws = torch.zeros(dll.ds_sched_workspace_bytes() // 4, dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
assert dll.ds_sched_set_workspace(_sched_ws.data_ptr(), _sched_ws.numel() * 4) == 0
dev = torch.device("cuda", 0)
B = 768
P = ctypes.c_void_p
st = P(torch.cuda.current_stream().cuda_stream)
LAYERS = [(80, 32, 64, 128, 5, 2, False, True), (40, 16, 128, 256, 5, 2, False, False), (20, 8, 256, 256, 3, 1, True, False),
          (20, 8, 256, 512, 5, 2, False, False), (10, 4, 512, 512, 3, 1, True, False)]
probe = torch.zeros(8 * 4 * 1024, dtype=torch.int64, device=dev)
dll.ds_f16_set_probe(P(probe.data_ptr()))
names = ["tile top: halo zero + stage write + barrier", "MFMA stream (all chunks)", "barrier after the stream", "epilogue"]
for (h, w, ci, co, k, s_, res, planes) in LAYERS:
    ho, wo = (h - 1) // s_ + 1, (w - 1) // s_ + 1
    x = torch.randn(B, h, w, ci, device=dev).abs().half()
    xin = x.view(B * h * w, ci // 16, 16).permute(1, 0, 2).contiguous() if planes else x
    wt = torch.randn(co, ci, k, k, device=dev) * (1.0 / (ci * k * k) ** 0.5)
    wp = torch.empty(wt.numel(), dtype=torch.float16, device=dev)
    assert dll.ds_pack_conv_weight_f16(P(wt.data_ptr()), P(wp.data_ptr()), co, ci, k, st) == 0
    sc, sh = torch.rand(co, device=dev) + 0.5, torch.randn(co, device=dev)
    r = (torch.randn(B, ho, wo, co, device=dev).abs() * 4).half() if res else None
    y = torch.empty(B, ho, wo, co, dtype=torch.float16, device=dev)
    shp = ConvShape(B, h, w, ci, co, k, s_)
    flags = DS_EPI_AFFINE | DS_EPI_CLIP | (DS_EPI_RESIDUAL if res else 0) | (DS_CONV_IN_PLANES16 if planes else 0)
    out8 = (ctypes.c_int * 8)()
    dll.ds_conv_f16_plan_describe_hinted(ctypes.byref(shp), flags, out8)
    args = (ctypes.byref(shp), P(xin.data_ptr()), P(wp.data_ptr()), P(sc.data_ptr()), P(sh.data_ptr()), P(r.data_ptr()) if res else None,
            P(y.data_ptr()), flags, st)
    for _ in range(3):
        assert dll.ds_conv_fwd_f16(*args) == 0
    torch.cuda.synchronize()
    probe.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    assert dll.ds_conv_fwd_f16(*args) == 0
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    t = probe.view(1024, 4, 8).cpu().numpy().astype(np.float64)
    used = t[:, 0, 0] > 0
    t = t[used]
    n_wg = t.shape[0]
    tiles_done = (t[:, :, 4] > 0).sum(1)
    print(f"conv{k}x{k}s{s_} {ci}->{co} {ho}x{wo}: {us:6.0f} us, {n_wg} workgroups, tile {out8[0]}x{out8[1]}, {out8[4]} tiles, plan {out8[7]}; "
          f"tiles per workgroup (recorded, <= 4): {np.bincount(tiles_done.astype(int)).tolist()}")
    # steady-state tiles: every recorded tile but a workgroup's first (its prologue ran before stamp 0 of tile 0)
    for label, sel in (("first tile of a workgroup", [0]), ("later tiles", [1, 2, 3])):
        rows = [t[i, j] for i in range(n_wg) for j in sel if t[i, j, 4] > 0]
        if not rows:
            continue
        a = np.array(rows)
        ph = np.diff(a[:, :5], axis=1)
        med = np.median(ph, axis=0)
        tot = np.median(a[:, 4] - a[:, 0])
        print(f"   {label}: " + " | ".join(f"{nm} {m:.0f} ({m / tot:.0%})" for nm, m in zip(names, med)) + f" | tile {tot:.0f} ticks")
    # s_memtime counters have bases of their own per clock domain (XCD): workgroups are clustered by their start stamps
    # (domains are millions of ticks apart, a launch is a few hundred thousand long); skews are taken inside each cluster
    ends = np.array([t[i, int(tiles_done[i]) - 1, 4] if tiles_done[i] > 0 else np.nan for i in range(n_wg)])
    starts = t[:, 0, 0]
    order = np.argsort(starts)
    cuts = np.nonzero(np.diff(starts[order]) > 2e6)[0] + 1
    rows = []
    for grp in np.split(order, cuts):
        m = grp[np.isfinite(ends[grp])]
        if len(m) < 4:
            continue
        s0, e0 = starts[m], ends[m]
        rows.append((e0.max() - s0.min(), s0.max() - s0.min(), e0.max() - e0.min(), np.median(e0 - s0), len(m),
                     np.percentile(s0 - s0.min(), 90), np.percentile(e0.max() - e0, 90)))
    r = np.array(rows)
    span = np.median(r[:, 0])
    print(f"   {len(rows)} clock domains of {sorted(set(int(v) for v in r[:, 4]))} workgroups; medians over them: first stamp -> last stamp {span:.0f} ticks = "
          f"{span / us:.0f} ticks/us of the event-timed launch; workgroup START skew {np.median(r[:, 1]):.0f} ({np.median(r[:, 1]) / span:.0%}; 90 % of them "
          f"within {np.median(r[:, 5]):.0f}), END skew {np.median(r[:, 2]):.0f} ({np.median(r[:, 2]) / span:.0%}; 90 % within {np.median(r[:, 6]):.0f} of the last), "
          f"a workgroup's own first-stamp -> last-stamp {np.median(r[:, 3]):.0f} ({np.median(r[:, 3]) / span:.0%})")
