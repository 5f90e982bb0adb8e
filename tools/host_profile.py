"""Host-side cost of one bench step (N = 1): where the Python / ctypes / torch time of the enqueue goes.

    python tools/host_profile.py [--steps 200] [--events]

Prints the per-phase host time (forward, filter + refinement, loss, search) and the cProfile top of the step.  The device
is kept busy but never waited for inside the measured loops (a synchronize every 20 steps keeps the queue bounded)."""
import argparse
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--events", action="store_true", help="with the per-launch event instrumentation of the bench")
    args = ap.parse_args()
    from deepspeaker_pytorch_amd.mining import mine_semihard_negatives, select_triplets
    from deepspeaker_pytorch_amd.model import DeepSpeakerModel, TripletMarginLoss, get_engine
    from deepspeaker_pytorch_amd.synthetic import synthetic_state_dict
    dev = torch.device("cuda", 0)
    sd = synthetic_state_dict(seed=0, num_classes=1211)
    model = DeepSpeakerModel(512, 1211, precision="f16")
    model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    model = model.to(dev).eval()
    g = torch.Generator(device="cpu").manual_seed(1234)
    data_all = torch.randn(768, 1, 160, 64, generator=g).to(dev)
    data = list(data_all.split(256))
    c1 = torch.randint(0, 64, (256,), generator=g)
    c2 = (c1 + 1 + torch.randint(0, 63, (256,), generator=g)) % 64
    c1, c2 = c1.to(dev), c2.to(dev)
    labels = torch.cat([c1, c1, c2])
    loss_fn = TripletMarginLoss(0.1)
    eng = get_engine()
    phases = {"forward": 0.0, "filter": 0.0, "loss": 0.0, "search": 0.0}

    def step(timed=False):
        with torch.no_grad():
            t0 = time.perf_counter()
            e_all = model(data_all)
            embs = list(e_all.split(256))
            t1 = time.perf_counter()
            sel = select_triplets(*embs, margin=0.1, model=model, inputs=data)
            t2 = time.perf_counter()
            loss = loss_fn.forward(*embs)
            t3 = time.perf_counter()
            mined = mine_semihard_negatives(embs[0], embs[1], c1, e_all, labels, side_stream=True)
            t4 = time.perf_counter()
        if timed:
            phases["forward"] += t1 - t0
            phases["filter"] += t2 - t1
            phases["loss"] += t3 - t2
            phases["search"] += t4 - t3
        return loss, sel, mined

    keep = []
    for i in range(40):
        keep.append(step())
    torch.cuda.synchronize()
    eng.profile = [] if args.events else None
    keep.clear()
    t0 = time.perf_counter()
    for i in range(args.steps):
        keep.append(step(True))
        if len(keep) > 40:
            del keep[:20]
        if i % 20 == 19:
            t_s = time.perf_counter()
            torch.cuda.synchronize()
            t0 += time.perf_counter() - t_s          # the wait is not host enqueue time
            if args.events:
                eng.profile = []
    total = time.perf_counter() - t0
    print(f"host enqueue {total / args.steps * 1e3:.3f} ms/step; phases (ms/step): "
          + ", ".join(f"{k} {v / args.steps * 1e3:.3f}" for k, v in phases.items()))
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for i in range(100):
        keep.append(step())
        if len(keep) > 40:
            del keep[:20]
        if i % 20 == 19:
            torch.cuda.synchronize()
            if args.events:
                eng.profile = []
    pr.disable()
    torch.cuda.synchronize()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(35)
    st.sort_stats("cumulative").print_stats(45)


if __name__ == "__main__":
    main()
