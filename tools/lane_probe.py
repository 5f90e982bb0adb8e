#!/usr/bin/env python3
"""Does the filter-gradient lane overlap with the caller's stream in THIS process?  Two spin kernels (torch.cuda._sleep, one
workgroup each) on the two streams: ~1x the spin = different hardware queues, ~2x = the same queue (they serialise).  Then
the fp16 and bf16x3 training steps of the bench, 20 steps after 30.  Run several fresh processes in a row:
    for i in 1 2 3 4 5 6; do python tools/lane_probe.py; done"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from deepspeaker_pytorch_amd import backward
from deepspeaker_pytorch_amd.model import DeepSpeakerModel, TripletMarginLoss
from deepspeaker_pytorch_amd.optim import create_optimizer
from deepspeaker_pytorch_amd.synthetic import synthetic_state_dict

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
SPIN = 2_000_000


def overlap(a, b):
    """elapsed of one spin on each of the two streams started together, in units of one spin alone"""
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(a):
        e0.record()
        torch.cuda._sleep(SPIN)
        e1.record()
    torch.cuda.synchronize()
    alone = e0.elapsed_time(e1)
    s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(a):
        s0.record()
        b.wait_event(s0)
        torch.cuda._sleep(SPIN)
    with torch.cuda.stream(b):
        torch.cuda._sleep(SPIN)
    a.wait_stream(b)
    with torch.cuda.stream(a):
        s1.record()
    torch.cuda.synchronize()
    return s0.elapsed_time(s1) / alone


tp = sys.argv[1] if len(sys.argv) > 1 else "f16"
sd = synthetic_state_dict(0, 1211)
m = DeepSpeakerModel(512, 1211, precision="f16", train_precision=None if tp == "bf16x3" else tp)
m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
m = m.to(dev).train()
opt = create_optimizer(m, 0.1, "adagrad", lr_decay=1e-4)
g = torch.Generator(device="cpu").manual_seed(1234)
data = list(torch.randn(768, 1, 160, 64, generator=g).to(dev).split(256))
loss_fn = TripletMarginLoss(0.1)


def step():
    out = m.forward_triplet(*data)
    loss = loss_fn.forward(*out)
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()


for _ in range(30):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    step()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 20 * 1e3
main = torch.cuda.current_stream(dev)
lane = backward._wgrad_streams.get((dev, 0))
others = [torch.cuda.Stream(device=dev) for _ in range(6)]
print(f"{tp}: {ms:.3f} ms/step; overlap(main, lane) = {overlap(main, lane):.2f}; fresh streams vs main: "
      + " ".join(f"{overlap(main, s):.2f}" for s in others) + "; vs lane: " + " ".join(f"{overlap(lane, s):.2f}" for s in others))
