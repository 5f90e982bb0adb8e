#!/usr/bin/env python3
"""Soak test: thousands of bench-style eval steps and hundreds of training steps in one process; prints the caching
allocator's allocated / reserved MiB along the way (leaks and cross-stream reuse stalls show up as growth).
python tools/soak.py [--no-overlap] [--eval-steps N] [--train-steps N]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--no-overlap", action="store_true", help="filter gradients on the main stream")
    ap.add_argument("--eval-steps", type=int, default=3000)
    ap.add_argument("--train-steps", type=int, default=300)
    args = ap.parse_args()
    from deepspeaker_pytorch_amd import backward
    from deepspeaker_pytorch_amd.mining import mine_semihard_negatives, select_triplets
    from deepspeaker_pytorch_amd.model import DeepSpeakerModel, TripletMarginLoss
    from deepspeaker_pytorch_amd.optim import create_optimizer
    from deepspeaker_pytorch_amd.synthetic import synthetic_state_dict
    backward.OVERLAP_FILTER_GRADIENTS = not args.no_overlap
    dev = torch.device("cuda", 0)
    sd = synthetic_state_dict(0, 1211)
    m = DeepSpeakerModel(512, 1211, precision="f16")
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    m = m.to(dev).eval()
    g = torch.Generator().manual_seed(1)
    data = [torch.randn(256, 1, 160, 64, generator=g).to(dev) for _ in range(3)]
    allx = torch.cat(data)
    lab = torch.randint(0, 64, (256,), generator=g).to(dev)
    labs = torch.cat([lab, lab, (lab + 1) % 64])
    loss_fn = TripletMarginLoss(0.1)

    def mem():
        return f"{torch.cuda.memory_allocated() >> 20} / {torch.cuda.memory_reserved() >> 20} MiB allocated / reserved"

    with torch.no_grad():
        for it in range(args.eval_steps):
            e = m(allx)
            ea, ep, en = e.split(256)
            sel = select_triplets(ea, ep, en, margin=0.1, model=m, inputs=data)
            loss = loss_fn.forward(ea, ep, en)
            mine_semihard_negatives(ea, ep, lab, e, labs, side_stream=True)
            if it in (100, args.eval_steps // 2, args.eval_steps - 1):
                torch.cuda.synchronize()
                print("eval step", it, mem(), "loss", float(loss), "selected", sel.n_selected, flush=True)
    m.train()
    opt = create_optimizer(m, 0.1)
    for it in range(args.train_steps):
        a, p, n = m.forward_triplet(*data)
        loss = loss_fn.forward(a, p, n)
        opt.zero_grad()
        loss.backward()
        opt.step()
        if it in (5, args.train_steps // 2, args.train_steps - 1):
            torch.cuda.synchronize()
            print("train step", it, mem(), "loss", float(loss.detach()), flush=True)
    print("soak ok")


if __name__ == "__main__":
    main()
