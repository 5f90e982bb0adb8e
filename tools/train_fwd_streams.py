#!/usr/bin/env python3
"""Probe: does the train-mode forward of a triplet step gain from running its members on separate streams (one member's
HBM-bound BatchNorm passes next to another member's convolutions) instead of in lock-step over one batch?
Prints ms per 768-utterance forward for: grouped (HEAD), three calls on one stream, three calls on 2 / 3 streams."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from deepspeaker_pytorch_amd.model import DeepSpeakerModel, get_engine
from deepspeaker_pytorch_amd.synthetic import synthetic_state_dict


def main():
    dev = torch.device("cuda", 0)
    sd = synthetic_state_dict(0, 1211)
    model = DeepSpeakerModel(512, 1211, precision="bf16x3")
    model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    model = model.to(dev).train()
    eng = get_engine()
    pw = model._packed(with_dgrad=True, with_bf16=True)
    bns = model._bn_params()
    x = torch.randn(768, 1, 160, 64, generator=torch.Generator().manual_seed(5)).to(dev)
    xs = [x[g * 256:(g + 1) * 256] for g in range(3)]
    streams = [torch.cuda.Stream() for _ in range(3)]

    def grouped():
        return eng.forward_train_group(xs, pw, bns, save=True, precision="bf16x3")

    def calls(n_streams):
        def run():
            if n_streams == 0:
                return [eng.forward_train(xg, pw, bns, save=True, precision="bf16x3") for xg in xs]
            cur = torch.cuda.current_stream()
            outs = []
            for g, xg in enumerate(xs):
                s = streams[g % n_streams]
                s.wait_stream(cur)
                with torch.cuda.stream(s):
                    outs.append(eng.forward_train(xg, pw, bns, save=True, precision="bf16x3"))
            for s in streams[:n_streams]:
                cur.wait_stream(s)
            return outs
        return run

    variants = [("grouped (HEAD)", grouped), ("3 calls, 1 stream", calls(0)), ("3 calls, 2 streams", calls(2)),
                ("3 calls, 3 streams", calls(3))]
    res = {n: [] for n, _ in variants}
    for rnd in range(8):
        for name, fn in variants:
            for _ in range(2 if rnd == 0 else 0):
                fn()
            torch.cuda.synchronize()
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            out = fn()
            t1.record()
            torch.cuda.synchronize()
            del out
            res[name].append(t0.elapsed_time(t1))
    for name, ts in res.items():
        print(f"{name:24s} median {np.median(ts):7.3f} ms   min {min(ts):7.3f}")


if __name__ == "__main__":
    main()
