#!/usr/bin/env python3
"""Where a persistent workgroup of the fused BasicBlock kernel spends a tile: a probe copy of the library (-DDS_F16_PROBE:
s_memtime stamps at the phase boundaries of each workgroup's first four tiles) runs the two bench blocks at B = 768 and
prints the per-phase medians in clocks of the steady-state tiles (tile 1..3 of every workgroup).
    build container: python tools/block_phase_probe.py --build-only [name -Dflag ...] ;  GPU box: ... [name]
Ablation builds (results wrong by construction, timing only): -DDS_ABL_NO_REFILL / -DDS_ABL_NO_AFRAG / -DDS_ABL_NO_STAGE."""
import ctypes
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CSRC = os.path.join(ROOT, "deepspeaker-pytorch_amd", "csrc")
ARGS = [a for a in sys.argv[1:] if a != "--build-only"]
NAME = ARGS[0] if ARGS else "base"             # variant name; further arguments: extra -D flags (ablations)
OUT = os.path.join(ROOT, "tools", "_ab", f"libds_blockprobe_{NAME}.so")


def build():
    objd = os.path.join(ROOT, "tools", "_ab", f"obj_blockprobe_{NAME}")
    os.makedirs(objd, exist_ok=True)
    srcs = [os.path.join(CSRC, f) for f in ("conv_block_f16.hip", "bn_pack.hip")]   # (bn_pack: the scheduling slots); filters are packed by the stock library

    def cc(s):
        o = os.path.join(objd, os.path.basename(s)[:-4] + ".o")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DDS_F16_PROBE", *ARGS[1:], "-mllvm",
                        "-pragma-unroll-threshold=1000000", "-Wno-pass-failed", f"-I{CSRC}", f"-I{ROOT}/include", "-c", "-o", o, s],
                       check=True)
        return o

    with ThreadPoolExecutor(16) as ex:
        objs = list(ex.map(cc, srcs))
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT, *objs], check=True)


if "--build-only" in sys.argv:
    build()
    print(OUT)
    sys.exit(0)

import numpy as np
import torch

dll = ctypes.CDLL(OUT)
# the probe copy of the library draws its tile counters from a zeroed buffer of ours, like the product wrapper does
dll.ds_sched_workspace_bytes.restype = ctypes.c_size_t
dll.ds_sched_set_workspace.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
_sched_ws = torch.zeros(dll.ds_sched_workspace_bytes() // 4, dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
assert dll.ds_sched_set_workspace(_sched_ws.data_ptr(), _sched_ws.numel() * 4) == 0
stock = ctypes.CDLL(os.path.join(ROOT, "deepspeaker-pytorch_amd", "libdeepspeaker_hip.so"))
dev = torch.device("cuda", 0)
B = 768
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = ctypes.c_void_p
names = ["halo + stage write + barrier", "first layer MFMA stream", "hand-over (barrier, bn1+clip -> LDS, barrier)",
         "second layer MFMA stream", "barrier", "epilogue"]
for (h, w, c) in ((80, 32, 64), (40, 16, 128)):
    x = torch.randn(B, h, w, c, device=dev).abs().half()
    packs = []
    for _ in range(2):
        wt = torch.randn(c, c, 3, 3, device=dev) * 0.05
        wp = torch.empty(wt.numel(), dtype=torch.float16, device=dev)
        stock.ds_pack_conv_weight_f16(P(wt.data_ptr()), P(wp.data_ptr()), c, c, 3, st)
        packs.append(wp)
    sc, sh = torch.ones(c, device=dev), torch.zeros(c, device=dev)
    y = torch.empty_like(x)
    n_wg = 512
    probe = torch.zeros(n_wg * 4 * 8, dtype=torch.int64, device=dev)
    dll.ds_block_set_probe(P(probe.data_ptr()))
    args = (P(x.data_ptr()), P(packs[0].data_ptr()), P(packs[1].data_ptr()), P(sc.data_ptr()), P(sh.data_ptr()), P(sc.data_ptr()),
            P(sh.data_ptr()), P(y.data_ptr()), B, h, w, c, 0, st)
    for _ in range(3):
        assert dll.ds_conv_block_f16(*args) == 0
    probe.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    assert dll.ds_conv_block_f16(*args) == 0
    e1.record()
    torch.cuda.synchronize()
    t = probe.view(n_wg, 4, 8).cpu().numpy().astype(np.float64)
    steady = t[:, 1:4, :7].reshape(-1, 7)
    steady = steady[steady[:, 0] > 0]
    ph = np.diff(steady, axis=1)
    first = np.diff(t[:, 0, :7], axis=1)
    gap = t[:, 1:4, 0] - t[:, 0:3, 6]
    print(f"block {c}ch {h}x{w}: {e0.elapsed_time(e1) * 1e3:.0f} us, tiles per workgroup {B * ((h + 7) // 8) / n_wg:.1f}")
    for nm, m, f in zip(names, np.median(ph, axis=0), np.median(first, axis=0)):
        print(f"   {nm:48s} {m:8.0f}   (first tile {f:8.0f})")
    print(f"   {'tile total':48s} {np.median(steady[:, 6] - steady[:, 0]):8.0f}   (first tile {np.median(t[:, 0, 6] - t[:, 0, 0]):8.0f}); "
          f"between tiles {np.median(gap):.0f}")
