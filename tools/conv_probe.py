"""Per-layer timing of the bf16x3 convolution through the C ABI, for one or more builds of the library.

    python tools/conv_probe.py [lib.so ...]          (default: the in-tree build)

Prints microseconds and algorithmic TFLOP/s per ResCNN layer shape at B=768 (BASELINE configs[1]).
Used for A/B runs of kernel variants inside ONE gpurun call (box-to-box variance is larger than most
kernel changes)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from deepspeaker_pytorch_amd import _native  # noqa: E402

LAYERS = [  # name, H, W, Cin, Cout, KS, stride
    ("s1 3x3 64", 80, 32, 64, 64, 3, 1),
    ("s2 5x5 64>128", 80, 32, 64, 128, 5, 2),
    ("s2 3x3 128", 40, 16, 128, 128, 3, 1),
    ("s3 5x5 128>256", 40, 16, 128, 256, 5, 2),
    ("s3 3x3 256", 20, 8, 256, 256, 3, 1),
    ("s4 5x5 256>512", 20, 8, 256, 512, 5, 2),
    ("s4 3x3 512", 10, 4, 512, 512, 3, 1),
]


def probe(path, B=768, reps=20):
    lib = _native.NativeLib(path)
    dev = torch.device("cuda:0")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    out = []
    for name, H, W, Cin, Cout, KS, s in LAYERS:
        g = torch.Generator(device="cpu").manual_seed(1)
        x = torch.rand(B, H, W, Cin, generator=g).to(dev)
        w = (torch.randn(Cout, Cin, KS, KS, generator=g) * 0.05).to(dev)
        n = Cout * Cin * KS * KS
        whi = torch.empty(n, dtype=torch.bfloat16, device=dev)
        wlo = torch.empty(n, dtype=torch.bfloat16, device=dev)
        lib.call("ds_pack_conv_weight_bf16", p(w), p(whi), p(wlo), Cout, Cin, KS, st)
        Ho, Wo = (H + 2 * (KS // 2) - KS) // s + 1, (W + 2 * (KS // 2) - KS) // s + 1
        y = torch.empty(B, Ho, Wo, Cout, device=dev)
        res = torch.rand(B, Ho, Wo, Cout, generator=g).to(dev)
        sc = torch.ones(Cout, device=dev)
        sh = torch.zeros(Cout, device=dev)
        shp = _native.ConvShape(B, H, W, Cin, Cout, KS, s)
        flags = 1 | 4 | (2 if s == 1 else 0)

        def run():
            lib.call("ds_conv_fwd_bf16", ctypes.byref(shp), p(x), p(whi), p(wlo), p(sc), p(sh),
                     p(res) if s == 1 else None, p(y), None, flags, st)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        fl = 2.0 * B * Ho * Wo * Cout * Cin * KS * KS
        out.append((name, us, fl / us * 1e-6, float(y.double().sum())))
    # conv1 (Cin = 1, VALU kernel)
    x = torch.rand(B, 160, 64, generator=torch.Generator().manual_seed(2)).to(dev)
    w = (torch.randn(64, 1, 5, 5, generator=torch.Generator().manual_seed(3)) * 0.2).to(dev)
    wp = torch.empty(1600, device=dev)
    lib.call("ds_pack_conv1_weight_f32", p(w), p(wp), 64, st)
    y = torch.empty(B, 80, 32, 64, device=dev)
    sc, sh = torch.ones(64, device=dev), torch.zeros(64, device=dev)
    for fn in ("ds_conv5x5s2_c1_fwd_f32", "ds_conv5x5s2_c1_fwd_bf16"):
        run1 = lambda: lib.call(fn, p(x), p(wp), p(sc), p(sh), p(y), None, B, 160, 64, 64, 1 | 4, st)
        for _ in range(3):
            run1()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            run1()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        out.append(("conv1 " + fn[-4:].strip("_"), us, 2.0 * B * 80 * 32 * 64 * 25 / us * 1e-6, float(y.double().sum())))
    return out


if __name__ == "__main__":
    paths = sys.argv[1:] or [os.path.join(_native._HERE, _native.LIB_NAME)]
    res = {q: probe(q) for q in paths}
    for i, lay in enumerate(LAYERS + [("conv1 VALU f32",), ("conv1 MFMA bf16x3",)]):
        cells = "  ".join(f"{res[q][i][1]:8.1f}us {res[q][i][2]:6.1f}TF" for q in paths)
        print(f"{lay[0]:16s} {cells}")
    print(" " * 16, "  ".join(f"{sum(r[1] for r in res[q][:len(LAYERS)]):8.1f}us" + " " * 9 for q in paths))
    for q in paths:
        print(os.path.basename(q), "checksums", " ".join(f"{r[3]:.6e}" for r in res[q]))
