#!/usr/bin/env python3
"""Micro-benchmark of the first convolution (5x5 s2, 1 -> 64 channels, BN + clip fused) at the bench batch:
event-timed launches of the matrix-core kernel with fp16 and f32 output, checked against the exact VALU kernel.
python tools/conv1_probe.py [--batch 768]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=768)
    ap.add_argument("--frames", type=int, default=160)
    args = ap.parse_args()
    from deepspeaker_pytorch_amd.engine import DS_EPI_AFFINE, DS_EPI_CLIP, DS_EPI_OUT_F16
    from deepspeaker_pytorch_amd.model import get_engine
    eng = get_engine()
    dev = torch.device("cuda", 0)
    g = torch.Generator().manual_seed(3)
    B, T = args.batch, args.frames
    x = torch.randn(B, 1, T, 64, generator=g).to(dev)
    w = (torch.randn(64, 1, 5, 5, generator=g) * 0.2).to(dev)
    wp = torch.empty(25 * 64, dtype=torch.float32, device=dev)
    eng.lib.call("ds_pack_conv1_weight_f32", eng._p(w), eng._p(wp), 64, eng._stream(x))
    sc = (torch.rand(64, generator=g) + 0.5).to(dev)
    sh = (torch.randn(64, generator=g) * 0.5).to(dev)
    out = {}
    ref, _ = eng.conv1(x, wp, B, T, 64, sc, sh, DS_EPI_AFFINE | DS_EPI_CLIP, lowp=False)
    for name, fl in (("f16_out", DS_EPI_AFFINE | DS_EPI_CLIP | DS_EPI_OUT_F16), ("f32_out", DS_EPI_AFFINE | DS_EPI_CLIP)):
        y, _ = eng.conv1(x, wp, B, T, 64, sc, sh, fl, lowp=True)
        err = float((y.float() - ref).abs().max())
        for _ in range(5):
            eng.conv1(x, wp, B, T, 64, sc, sh, fl, lowp=True)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        n = 50
        ev[0].record()
        for _ in range(n):
            eng.conv1(x, wp, B, T, 64, sc, sh, fl, lowp=True)
        ev[1].record()
        torch.cuda.synchronize()
        us = ev[0].elapsed_time(ev[1]) / n * 1e3
        byts = x.numel() * 4 + y.numel() * y.element_size()
        out[name] = {"us": round(us, 1), "GB_per_s": round(byts / us / 1e3, 1), "max_abs_err_vs_valu_f32": err}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
