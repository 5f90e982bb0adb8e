#!/usr/bin/env python3
"""Secondary benchmark (not the headline): one training step as the reference runs it
(train_triplet.py:215-224): three train-mode forwards of 256 utterances, TripletMarginLoss, backward,
Adagrad step.  Prints one JSON line: utterances/s through forward+backward+update on one MI355X."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--precision", default="bf16x3", choices=["f32", "bf16x3"])
    ap.add_argument("--grouped", action="store_true", help="model.forward_triplet instead of three calls")
    ap.add_argument("--no-overlap", action="store_true", help="filter gradients on the main stream (A/B of backward._FilterGradLane)")
    ap.add_argument("--no-fuse-bn", action="store_true", help="3x3 data gradients and the BatchNorm backward below them as separate launches (A/B of backward._dgrad_bn_bwd)")
    ap.add_argument("--force-dp", action="store_true", help="every data-parallel branch with one rank (RCCL)")
    ap.add_argument("--no-dp-streams", action="store_true", help="data-parallel forward in lock-step (A/B of Engine.MEMBER_STREAMS_DP)")
    ap.add_argument("--cprofile", action="store_true", help="print the host-side profile of the timed steps")
    args = ap.parse_args()
    from deepspeaker_pytorch_amd.model import DeepSpeakerModel, TripletMarginLoss
    from deepspeaker_pytorch_amd import backward
    backward.OVERLAP_FILTER_GRADIENTS = not args.no_overlap
    backward.FUSE_DGRAD_BN_BWD = not args.no_fuse_bn
    from deepspeaker_pytorch_amd.synthetic import synthetic_state_dict
    dev = torch.device("cuda", 0)
    sd = synthetic_state_dict(0, 1211)
    model = DeepSpeakerModel(512, 1211, precision=args.precision)
    model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    model = model.to(dev).train()
    from deepspeaker_pytorch_amd.engine import Engine
    Engine.MEMBER_STREAMS_DP = not args.no_dp_streams
    if args.force_dp:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        model.enable_data_parallel(force=True)
    from deepspeaker_pytorch_amd.optim import create_optimizer
    opt = create_optimizer(model, 0.1)              # fused multi-tensor Adagrad (train_triplet.py:379-382)
    g = torch.Generator().manual_seed(5)
    data = [torch.randn(args.batch, 1, 160, 64, generator=g).to(dev) for _ in range(3)]
    loss_fn = TripletMarginLoss(0.1)

    def step():
        if args.grouped:
            out_a, out_p, out_n = model.forward_triplet(*data)
        else:
            out_a, out_p, out_n = model(data[0]), model(data[1]), model(data[2])
        loss = loss_fn.forward(out_a, out_p, out_n)
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    prof = None
    if args.cprofile:
        import cProfile
        prof = cProfile.Profile()
        prof.enable()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if prof is not None:
        import pstats
        prof.disable()
        pstats.Stats(prof).sort_stats("cumulative").print_stats(35)
    print("host enqueue ms/step:", round(t_enq / args.steps * 1e3, 2))
    utt = 3 * args.batch * args.steps
    print(json.dumps({"metric": "training utterances/sec (fwd + bwd + Adagrad), " + args.precision, "value": round(utt / dt, 1),
                      "ms_per_step": round(dt / args.steps * 1e3, 2), "batch_triplets": args.batch,
                      "tflops_algorithmic": round(utt / dt * 6.9e9 / 1e12, 1), "final_loss": float(loss)}))


if __name__ == "__main__":
    main()
