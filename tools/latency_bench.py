"""Serving latency of the eval forward at small batch: eager launches vs HIP-graph replay (DeepSpeakerModel.graphed),
with and without the split-K small-launch path of the fp16 kernel (DeepSpeakerModel(low_latency=True)).
python tools/latency_bench.py [--precision f16]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precision", default="f16", choices=["f32", "bf16x3", "f16"])
    args = ap.parse_args()
    from deepspeaker_pytorch_amd.model import DeepSpeakerModel
    from deepspeaker_pytorch_amd.synthetic import synthetic_state_dict
    dev = torch.device("cuda", 0)
    sd = synthetic_state_dict(0, 8)
    out = {}
    for low in ((False, True) if args.precision == "f16" else (False,)):
        model = DeepSpeakerModel(512, 8, precision=args.precision, low_latency=low)
        model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
        model = model.to(dev).eval()
        for b in (1, 4, 16, 64):
            x = torch.randn(b, 1, 160, 64, device=dev)
            g = model.graphed(x)
            with torch.no_grad():
                ref = model(x).clone()
                assert torch.equal(g(x), ref), "graph replay differs from the eager forward"

                def timeit(fn, n=200):
                    for _ in range(20):
                        fn()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(n):
                        fn()
                    torch.cuda.synchronize()
                    return (time.perf_counter() - t0) / n * 1e6
                out[f"B={b}" + (" split-K" if low else "")] = {"eager_us": round(timeit(lambda: model(x)), 1),
                                                                "graph_us": round(timeit(lambda: g(x)), 1)}
    print(json.dumps({"metric": "eval forward latency, 160-frame utterances, " + args.precision, **out}))


if __name__ == "__main__":
    main()
