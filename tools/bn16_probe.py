#!/usr/bin/env python3
"""Each element-wise kernel of the fp16 training step ALONE on the chip, at the four stages' tensor shapes of the 768-row
step: microseconds and effective HBM GB/s (algorithmic bytes).  python tools/bn16_probe.py   (DS_TF_GRID=<cap> to vary the
grid cap of the grid-stride passes)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from deepspeaker_pytorch_amd.model import get_engine


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    eng = get_engine()
    dev = torch.device("cuda", 0)
    G, Bm = 3, 256
    print(f"DS_TF_GRID={os.environ.get('DS_TF_GRID', '(8192)')}")
    for (h, w, c) in ((80, 32, 64), (40, 16, 128), (20, 8, 256), (10, 4, 512)):
        B = G * Bm
        n_pix = Bm * h * w
        n = B * h * w * c
        z = (torch.randn(B, h, w, c, device=dev) * 2).half()
        g1 = (torch.randn(B, h, w, c, device=dev) * 0.1).half()
        g2 = (torch.randn(B, h, w, c, device=dev) * 0.1).half()
        act = (torch.randn(B, h, w, c, device=dev) * 8 + 8).clamp(0, 20).half()
        y = torch.empty_like(z)
        gy, gz = torch.empty_like(z), torch.empty_like(z)
        tables = torch.rand(4, G, c, device=dev) + 0.5
        gamma = torch.rand(c, device=dev) + 0.5
        rows = eng.lib.raw("ds_bn_f16_partial_rows")(n_pix, c)
        partial = torch.empty(G, rows, c, 2, device=dev)
        coef = torch.empty(G, 3 * c, device=dev)
        gg, gb = torch.empty(c, device=dev), torch.empty(c, device=dev)
        rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
        beta = torch.zeros(c, device=dev)
        P = eng._p
        mb = n * 2 / 1e6                                        # MB per fp16 tensor pass

        def report(name, us, passes):
            print(f"  [{h}x{w}x{c}] {name:34s} {us:8.1f} us  {passes} passes  {passes * mb / us:6.2f} TB/s")

        report("stats (partials + fold)", timeit(lambda: eng.lib.call(
            "ds_bn_stats_group_f16", P(z), P(partial), n_pix, P(gamma), P(beta), 1e-5, 0.1, P(rm), P(rv), P(tables[0]), P(tables[1]),
            P(tables[2]), P(tables[3]), c, G, None)), 1)
        report("stats partials only", timeit(lambda: eng.lib.call("ds_bn_stats_partial_f16", P(z), P(partial), n_pix, c, G, None)), 1)
        report("apply (clip)", timeit(lambda: eng.lib.call(
            "ds_bn_apply_group_f16", P(z), P(tables[2]), P(tables[3]), None, P(y), n_pix, c, G, 4, None)), 2)
        report("apply (clip + residual)", timeit(lambda: eng.lib.call(
            "ds_bn_apply_group_f16", P(z), P(tables[2]), P(tables[3]), P(act), P(y), n_pix, c, G, 4 | 2, None)), 3)
        report("bwd reduce, mask from z, no gy", timeit(lambda: eng.lib.call(
            "ds_bn_bwd_group_reduce_f16", P(g1), 0, None, None, 0, P(tables[2]), P(tables[3]), P(z), P(tables[0]), P(tables[1]), None,
            P(partial), n_pix, h, w, c, G, None)), 2)
        report("bwd reduce, g2, mask z, gy", timeit(lambda: eng.lib.call(
            "ds_bn_bwd_group_reduce_f16", P(g1), 0, P(g2), None, 0, P(tables[2]), P(tables[3]), P(z), P(tables[0]), P(tables[1]), P(gy),
            P(partial), n_pix, h, w, c, G, None)), 4)
        report("bwd reduce, act mask, gy", timeit(lambda: eng.lib.call(
            "ds_bn_bwd_group_reduce_f16", P(g1), 0, None, P(act), 0, None, None, P(z), P(tables[0]), P(tables[1]), P(gy),
            P(partial), n_pix, h, w, c, G, None)), 4)
        report("bwd whole (reduce+fold+apply), gy", timeit(lambda: eng.lib.call(
            "ds_bn_bwd_group_f16", P(g1), 0, None, P(act), 0, None, None, P(z), P(tables[0]), P(tables[1]), P(gamma), P(gy), P(partial),
            P(coef), P(gg), P(gb), P(gz), n_pix, h, w, c, G, 1.0, None)), 7)
        report("bwd whole, mask z, regen (no gy)", timeit(lambda: eng.lib.call(
            "ds_bn_bwd_group_f16", P(g1), 0, None, None, 0, P(tables[2]), P(tables[3]), P(z), P(tables[0]), P(tables[1]), P(gamma), None,
            P(partial), P(coef), P(gg), P(gb), P(gz), n_pix, h, w, c, G, 1.0, None)), 5)
        t = torch.empty_like(z)
        report("torch copy_ (fp16 -> fp16)", timeit(lambda: t.copy_(z)), 2)


if __name__ == "__main__":
    main()
