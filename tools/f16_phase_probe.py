#!/usr/bin/env python3
"""Where a workgroup of the fp16 convolution spends its time: builds a probe copy of the library
(-DDS_F16_PROBE: s_memtime stamps at kernel entry / end of prologue / end of the MFMA stream / after the barrier /
end of the epilogue), runs each bench layer at B = 768 and prints the per-phase medians in shader-clock ticks
(s_memtime counts at 100 MHz on gfx950: 1 tick = 10 ns).  python tools/f16_phase_probe.py [extra hipcc flags]"""
import ctypes
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from deepspeaker_pytorch_amd._native import ConvShape, NativeLib, DS_EPI_AFFINE, DS_EPI_CLIP, DS_EPI_RESIDUAL

CSRC = os.path.join(ROOT, "deepspeaker-pytorch_amd", "csrc")
OUT = "/tmp/libds_probe.so"
srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DDS_F16_PROBE",
                "-mllvm", "-pragma-unroll-threshold=1000000", "-Wno-pass-failed", f"-I{CSRC}", f"-I{ROOT}/include",
                *sys.argv[1:], "-o", OUT, *[s for s in srcs if "f16" in s or "bn_pack" in s or "tail_loss" in s]],
               check=True)
dll = ctypes.CDLL(OUT)
dev = torch.device("cuda", 0)
LAYERS = [(80, 32, 64, 64, 3, 1, True), (80, 32, 64, 128, 5, 2, False), (40, 16, 128, 128, 3, 1, True),
          (40, 16, 128, 256, 5, 2, False), (20, 8, 256, 256, 3, 1, True), (20, 8, 256, 512, 5, 2, False),
          (10, 4, 512, 512, 3, 1, True)]
B = 768
probe = torch.zeros(8 * 65536, dtype=torch.int64, device=dev)
dll.ds_f16_set_probe(ctypes.c_void_p(probe.data_ptr()))
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for (h, w, ci, co, k, s_, res) in LAYERS:
    ho, wo = (h - 1) // s_ + 1, (w - 1) // s_ + 1
    x = torch.randn(B, h, w, ci, device=dev).abs().half()
    wt = torch.randn(co, ci, k, k, device=dev) * 0.05
    wp = torch.empty(wt.numel(), dtype=torch.float16, device=dev)
    dll.ds_pack_conv_weight_f16(ctypes.c_void_p(wt.data_ptr()), ctypes.c_void_p(wp.data_ptr()), co, ci, k, st)
    y = torch.empty(B, ho, wo, co, dtype=torch.float16, device=dev)
    r = torch.randn(B, ho, wo, co, device=dev).abs().half() if res else None
    sc, sh = torch.ones(co, device=dev), torch.zeros(co, device=dev)
    shp = ConvShape(B, h, w, ci, co, k, s_)
    out8 = (ctypes.c_int * 8)()
    dll.ds_conv_f16_plan_describe(ctypes.byref(shp), out8)
    flags = DS_EPI_AFFINE | DS_EPI_CLIP | (DS_EPI_RESIDUAL if res else 0)
    args = (ctypes.byref(shp), ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(wp.data_ptr()), ctypes.c_void_p(sc.data_ptr()),
            ctypes.c_void_p(sh.data_ptr()), ctypes.c_void_p(r.data_ptr()) if res else None, ctypes.c_void_p(y.data_ptr()),
            flags, st)
    for _ in range(3):
        assert dll.ds_conv_fwd_f16(*args) == 0
    probe.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    assert dll.ds_conv_fwd_f16(*args) == 0
    e1.record()
    torch.cuda.synchronize()
    n = out8[4]
    t = probe[:8 * n].view(n, 8).cpu().numpy().astype(np.float64)
    ph = np.diff(t[:, :5], axis=1)
    med = np.median(ph, axis=0)
    fine = [np.median(t[:, 5] - t[:, 0]), np.median(t[:, 6] - t[:, 5]), np.median(t[:, 7] - t[:, 6]), np.median(t[:, 1] - t[:, 7])]
    span = (t[:, 4].max() - t[:, 0].min())
    print(f"conv{k}x{k}s{s_} {ci}->{co} {ho}x{wo}: {e0.elapsed_time(e1) * 1e3:6.0f} us, {n} wgs, tile {out8[0]}x{out8[1]} db/nit {out8[7]}; "
          f"ticks(10ns) prologue {med[0]:.0f} | mfma stream {med[1]:.0f} | barrier {med[2]:.0f} | epilogue {med[3]:.0f} | "
          f"wg total {np.median(t[:, 4] - t[:, 0]):.0f}")
    print(f"      prologue = ring preload + segment table {fine[0]:.0f} / barrier + descriptors {fine[1]:.0f} / "
          f"load issue + zero fill + row table {fine[2]:.0f} / fragment offsets {fine[3]:.0f}")
