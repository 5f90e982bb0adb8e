#!/usr/bin/env python3
"""Would a ONE-MFMA fp16 training step (fp16 operands for the forward and the data gradient, fp16 activations in HBM,
f32 accumulation, f32-class filter gradients) hold the gradient bars of tests/test_gpu_train_parity.py?  CPU simulation
on the reference's own ATen convolutions (oracle/torch_restatement.py): every convolution operand is rounded the way the
kernel would round it, the backward of a convolution is taken with rounded operands too (relative rounding of the
incoming gradient: what loss scaling achieves), and the result is compared with the float64 step evaluated with THE SAME
clip masks (the masked oracle of the GPU test).  Prints the worst relative L2 error over the 38 gradient tensors for
f32, split-bf16 (3 products), fp16 (1 product) and bf16 (1 product).     python tools/f16_train_sim.py [rows per member]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import torch
import torch.nn.functional as F
import torch.nn.grad as G

import deepspeaker_oracle as O
import torch_restatement as TR

torch.set_num_threads(8)


def rnd(t, mant):
    """round to `mant` significand bits (relative rounding, no range limits): 11 = fp16, 8 = bf16"""
    if mant is None:
        return t
    m, e = torch.frexp(t.double())
    return torch.ldexp(torch.round(m * 2.0 ** mant) / 2.0 ** mant, e).to(t.dtype)


def split3(t):
    """bf16 hi + lo: 16 significand bits carried by three products"""
    return rnd(t, 16)


class QConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, stride, pad, q):
        xq, wq = q(x), q(w)
        ctx.save_for_backward(xq, wq)
        ctx.cfg = (stride, pad, q)
        return F.conv2d(xq, wq, None, stride, pad)

    @staticmethod
    def backward(ctx, g):
        xq, wq = ctx.saved_tensors
        stride, pad, q = ctx.cfg
        gx = G.conv2d_input(xq.shape, wq, q(g), stride, pad)          # data gradient on the matrix cores: rounded operands
        gw = G.conv2d_weight(xq, wq.shape, g, stride, pad)           # filter gradient: f32-class arithmetic on the saved (rounded) input
        return gx, gw, None, None, None


def step(sd, xs, q, masks=None, dtype=torch.float32, act_q=None):
    names = [k for k in sd if ("running" not in k and "num_batches" not in k and "classifier" not in k)]
    params = {k: sd[k].detach().to(dtype).clone().requires_grad_(True) for k in names}
    embs, acts = [], []
    for g_, x in enumerate(xs):
        taps = {}
        x = x.to(dtype)
        mk = None if masks is None else masks[g_]

        def bn(t, name):
            return F.batch_norm(t, None, None, params[name + ".weight"], params[name + ".bias"], True, 0.1, 1e-5)

        def clip(t, key):
            y = F.hardtanh(t, 0.0, 20.0) if mk is None else TR._ClipFixedMask.apply(t, mk[key])
            if act_q is not None:
                y = y + (act_q(y) - y).detach()                      # stored rounded (straight-through)
            taps[key] = y.detach()
            return y

        for i in range(1, 5):
            x = QConv.apply(x, params[f"model.conv{i}.weight"], 2, 2, q) if i > 1 else F.conv2d(x, params["model.conv1.weight"], None, 2, 2)
            x = clip(bn(x, f"model.bn{i}"), f"stage{i}.a")
            r = x
            y = QConv.apply(x, params[f"model.layer{i}.0.conv1.weight"], 1, 1, q)
            y = clip(bn(y, f"model.layer{i}.0.bn1"), f"stage{i}.b")
            y = QConv.apply(y, params[f"model.layer{i}.0.conv2.weight"], 1, 1, q)
            y = bn(y, f"model.layer{i}.0.bn2")
            x = clip(y + r, f"stage{i}.c")
        x = F.adaptive_avg_pool2d(x, (1, None)).view(x.size(0), -1)
        x = F.linear(x, params["model.fc.weight"], params["model.fc.bias"])
        embs.append(x / torch.sqrt(torch.sum(x * x, 1) + 1e-10).view(-1, 1) * 10)
        acts.append(taps)
    loss = TR.triplet_loss(embs[0], embs[1], embs[2], 0.1)
    # backward from FIXED embedding gradients: a hinge that flips between two arithmetics (one of a handful of triplets)
    # would swamp what is measured here, the network's backward
    sum((e * g).sum() for e, g in zip(embs, GE[:len(embs)])).backward()
    return loss.detach(), embs, {k: v.grad for k, v in params.items()}, acts


GE = []


def main():
    bm = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    GE[:] = [torch.from_numpy(np.random.RandomState(70 + i).randn(bm, 512).astype(np.float32)) for i in range(3)]
    sd = {k: torch.from_numpy(np.array(v)) for k, v in O.make_state_dict(seed=31, num_classes=16).items()}
    xs = [torch.from_numpy(O.make_input(seed=32 + i, batch=bm)) for i in range(3)]
    modes = [("f32 operands", lambda t: t, None), ("bf16 x 3 (16 bits)", split3, None),
             ("fp16 x 1, fp16 activations", lambda t: rnd(t, 11), lambda t: rnd(t, 11)),
             ("fp16 x 1, f32 activations", lambda t: rnd(t, 11), None), ("bf16 x 1", lambda t: rnd(t, 8), None)]
    for name, q, aq in modes:
        loss, embs, grads, acts = step(sd, xs, q, act_q=aq)
        masks = [{k: (a > 0) & (a < 20) for k, a in d.items()} for d in acts]
        ref = TR.triplet_train_step(sd, xs, 0.1, masks=masks, dtype=torch.float64, ge=GE)
        worst = max((float((grads[k].double() - ref["grads"][k]).norm() / ref["grads"][k].norm()), k) for k in grads)
        med = float(np.median([float((grads[k].double() - ref["grads"][k]).norm() / ref["grads"][k].norm()) for k in grads]))
        e = max(float((a.double() - b).abs().max() / b.abs().max()) for a, b in zip(embs, ref["embeddings"]))
        print(f"{name:30s} embeddings {e:.1e}  "
              f"gradients rel-L2: median {med:.1e}, worst {worst[0]:.1e} ({worst[1]})")


if __name__ == "__main__":
    main()
