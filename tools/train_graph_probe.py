#!/usr/bin/env python3
"""Can one whole training step (train-mode forward of a / p / n, triplet loss, backward on two streams, fused optimizer) be
captured into ONE HIP graph and replayed?  Probe for the round-5 review's item 5.
    python tools/train_graph_probe.py [bf16x3|f16] [sgd|adagrad]
Prints eager vs replay ms per step (3 regions of 20 steps each) and the loss trajectories of both."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from deepspeaker_pytorch_amd.model import DeepSpeakerModel, TripletMarginLoss
from deepspeaker_pytorch_amd.optim import create_optimizer
from deepspeaker_pytorch_amd.synthetic import synthetic_state_dict

tp = sys.argv[1] if len(sys.argv) > 1 else "f16"
optn = sys.argv[2] if len(sys.argv) > 2 else "sgd"
dev = torch.device("cuda", 0)
sd = synthetic_state_dict(seed=0, num_classes=1211)
g = torch.Generator(device="cpu").manual_seed(1234)
xs = [torch.randn(256, 1, 160, 64, generator=g).to(dev) for _ in range(3)]
loss_fn = TripletMarginLoss(0.1)


def build():
    kw = dict(precision="f16", train_precision="f16") if tp == "f16" else dict(precision=tp)
    m = DeepSpeakerModel(512, 1211, **kw)
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    m = m.to(dev).train()
    return m, create_optimizer(m, 0.01, optn, lr_decay=1e-4)


def make_step(m, opt, losses):
    def step():
        out = m.forward_triplet(*xs)
        loss = loss_fn.forward(*out)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        losses.append(loss.detach())
    return step


def regions(fn, n=3, k=20):
    out = []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k):
            fn()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / k * 1e3)
    return out


# eager
m, opt = build()
le = []
step = make_step(m, opt, le)
for _ in range(10):
    step()
if "--no-eager-timing" not in sys.argv:
    print(f"[{tp} {optn}] eager ms/step:", [round(v, 3) for v in regions(step)])
else:
    for _ in range(4):
        step()
eager_losses = [float(v) for v in le[:14]]

# graph
m, opt = build()
lg = []
step = make_step(m, opt, lg)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
static_loss = []
try:
    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
        step_losses = []
        make_step(m, opt, step_losses)()
        static_loss.append(step_losses[0])
except Exception as exc:
    import traceback
    print("CAPTURE FAILED:", type(exc).__name__, str(exc)[:300])
    tb = traceback.format_exc().splitlines()
    print("\n".join(l for l in tb if "File" in l or "Error" in l)[-3000:])
    sys.exit(1)
torch.cuda.synchronize()
traj = [float(v) for v in lg[:3]]
for _ in range(11):
    graph.replay()
    traj.append(float(static_loss[0]))
print(f"[{tp} {optn}] graph replay ms/step:", [round(v, 3) for v in regions(graph.replay)])
print("eager losses :", " ".join(f"{v:.5f}" for v in eager_losses))
print("graph losses :", " ".join(f"{v:.5f}" for v in traj))
