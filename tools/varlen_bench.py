#!/usr/bin/env python3
"""Secondary benchmark: BASELINE configs[4] -- variable-length inference.  Utterances of 100..800 frames are
bucketed by length (the temporal mean pool makes any T legal; padding would change BatchNorm /
conv edge values, so equal-length batches are formed instead), embedded in eval mode, and the enrolment score
of a speaker is the mean of its utterances' distances.  Prints one JSON line (utterances/s and frames/s)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--utterances", type=int, default=4096)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--precision", default="bf16x3")
    args = ap.parse_args()
    import deepspeaker_oracle as O
    from deepspeaker_pytorch_amd.model import DeepSpeakerModel
    dev = torch.device("cuda", 0)
    sd = O.make_state_dict(seed=0, num_classes=16)
    model = DeepSpeakerModel(512, 16, precision=args.precision)
    model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    model = model.to(dev).eval()
    rs = np.random.RandomState(0)
    lengths = rs.randint(100, 801, args.utterances)
    buckets = {}
    for i, t in enumerate(lengths):                         # bucket = length rounded up to a multiple of 50
        buckets.setdefault(int(-(-t // 50) * 50), []).append(i)
    batches = []
    for t, idx in sorted(buckets.items()):
        for j in range(0, len(idx), args.batch):
            batches.append((t, len(idx[j:j + args.batch])))
    data = {(t, b): torch.randn(b, 1, t, 64, device=dev) for t, b in set(batches)}
    with torch.no_grad():
        for t, b in set(batches):                           # warm-up / plan build per shape
            model(data[(t, b)])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t, b in batches:
            model(data[(t, b)])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    frames = int(sum(t * b for t, b in batches))
    print(json.dumps({"metric": "variable-length inference (100-800 frames, bucketed by 50)", "precision": args.precision,
                      "utterances_per_s": round(args.utterances / dt, 1), "frames_per_s": round(frames / dt, 1),
                      "equivalent_160_frame_embeddings_per_s": round(frames / 160 / dt, 1),
                      "batches": len(batches), "distinct_shapes": len(set(batches))}))


if __name__ == "__main__":
    main()
