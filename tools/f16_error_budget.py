"""Where the fp16 eval path's embedding error comes from on a TRAINED network (VERDICT r4 "next" #2) -- CPU simulation.

Trains the fixture of tests/test_gpu_offdist.py (40 SGD steps, lr 0.02, momentum 0.9 / dampening 0.9, 32 triplets of a
16-speaker corpus of std-12 features) with the torch restatement on the host, then evaluates the eval forward with the
stage convolutions' operands rounded to fp16 -- everywhere, in one layer only, and everywhere but one layer; weights
only / activations only -- and prints each variant's embedding error (max |d| / max |ref|, the tests' measure).
Candidate fixes are evaluated the same way by tools/f16_error_fixes.py.

    python tools/f16_error_budget.py [--steps 40] [--opt sgd|adagrad|init] [--cache /tmp/f16_budget.pt]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import deepspeaker_oracle as O           # noqa: E402
import torch_restatement as TR           # noqa: E402

ALL = []          # the eleven stage convolutions, in forward order (conv1 runs split-operand bf16 in every mode)
for i in range(1, 5):
    if i > 1:
        ALL.append(f"model.conv{i}")
    ALL += [f"model.layer{i}.0.conv1", f"model.layer{i}.0.conv2"]


def train(opt_name, steps, lr, triplets=32, speakers=16, utts=8):
    sd = {k: torch.from_numpy(np.array(v)) for k, v in O.make_state_dict(seed=5, num_classes=speakers, randomize_bn=False).items()}
    corpus = O.make_speaker_corpus(31, speakers, utts, 160, scale=12.0, mix=(0.4, 0.3, 0.85))
    if opt_name == "init":
        return sd, corpus
    names = [k for k in sd if ("running" not in k and "num_batches" not in k and "classifier" not in k)]
    params = {k: sd[k].clone().requires_grad_(True) for k in names}
    running = {k: sd[k].clone() for k in sd if "running" in k}
    if opt_name == "sgd":
        opt = torch.optim.SGD(list(params.values()), lr=lr, momentum=0.9, dampening=0.9)
    else:
        opt = torch.optim.Adagrad(list(params.values()), lr=lr, lr_decay=1e-4)
    for it in range(steps):
        t0 = time.time()
        a, p, n, _, _ = O.sample_triplets(500 + it, speakers, utts, triplets)
        xs = [torch.from_numpy(O.gather_utterances(corpus, i)) for i in (a, p, n)]
        embs = [TR.forward_train(params, x, running) for x in xs]
        loss = TR.triplet_loss(*embs, 0.1)
        opt.zero_grad()
        loss.backward()
        opt.step()
        print(f"step {it}: loss {float(loss):.4f} ({time.time() - t0:.1f}s)", flush=True)
    out = dict(sd)
    out.update({k: v.detach() for k, v in params.items()})
    out.update(running)
    return out, corpus


def h(t):
    return t.half().float()


def forward(sd, x, w_of, act_round, store_round):
    """Eval forward; w_of(name) -> the filter to use for layer `name`; act_round(name, t) rounds the input of layer
    `name`; store_round(key, t) rounds a stored activation (the residual reads it too)."""
    def bn(t, name):
        return F.batch_norm(t, sd[name + ".running_mean"], sd[name + ".running_var"], sd[name + ".weight"],
                            sd[name + ".bias"], False, 0.1, 1e-5)
    for i in range(1, 5):
        nm = f"model.conv{i}"
        if i == 1:
            x = F.conv2d(x, sd[nm + ".weight"], None, 2, 2)
        else:
            x = F.conv2d(act_round(nm, x), w_of(nm), None, 2, 2)
        x = store_round(f"s{i}a", F.hardtanh(bn(x, f"model.bn{i}"), 0.0, 20.0))
        r = x
        nm = f"model.layer{i}.0.conv1"
        y = F.conv2d(act_round(nm, x), w_of(nm), None, 1, 1)
        y = store_round(f"s{i}b", F.hardtanh(bn(y, f"model.layer{i}.0.bn1"), 0.0, 20.0))
        nm = f"model.layer{i}.0.conv2"
        y = F.conv2d(act_round(nm, y), w_of(nm), None, 1, 1)
        y = bn(y, f"model.layer{i}.0.bn2")
        x = F.hardtanh(y + r, 0.0, 20.0)
        if i < 4:
            x = store_round(f"s{i}c", x)
    x = F.adaptive_avg_pool2d(x, (1, None)).reshape(x.size(0), -1)
    x = F.linear(x, sd["model.fc.weight"], sd["model.fc.bias"])
    norm = torch.sqrt(torch.sum(x * x, 1) + 1e-10)
    return x / norm.view(-1, 1) * 10


def run(sd, x, w_layers, a_layers, wfix=None):
    """fp16 filters on w_layers, fp16 stored activations feeding a_layers (a stored tensor is rounded once; it is
    rounded if ANY of its consumers is in a_layers -- as the product path stores it)."""
    wcache = {}

    def w_of(nm):
        if nm not in wcache:
            w = sd[nm + ".weight"]
            if nm in w_layers:
                w = wfix(nm, w) if wfix is not None else h(w)
            wcache[nm] = w
        return wcache[nm]
    consumer = {}
    for i in range(1, 5):
        consumer[f"s{i}a"] = f"model.layer{i}.0.conv1"
        consumer[f"s{i}b"] = f"model.layer{i}.0.conv2"
        if i < 4:
            consumer[f"s{i}c"] = f"model.conv{i + 1}"

    def store_round(key, t):
        return h(t) if consumer[key] in a_layers else t
    with torch.no_grad():
        return torch.cat([forward(sd, x[i:i + 64], w_of, lambda nm, t: t, store_round) for i in range(0, x.shape[0], 64)])


def err(e, ref):
    return float((e - ref).abs().max() / ref.abs().max())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--opt", default="sgd")
    ap.add_argument("--lr", type=float, default=0.02)
    ap.add_argument("--cache", default="/tmp/f16_budget_{opt}.pt")
    ap.add_argument("--rows", type=int, default=768)
    ap.add_argument("--quick", action="store_true", help="skip the per-layer tables")
    args = ap.parse_args()
    torch.set_num_threads(os.cpu_count() or 8)
    cache = args.cache.format(opt=args.opt)
    if os.path.exists(cache):
        sd, corpus = torch.load(cache, weights_only=False)
    else:
        sd, corpus = train(args.opt, args.steps, args.lr)
        torch.save((sd, corpus), cache)
    a, p, n, _, _ = O.sample_triplets(9000, corpus.shape[0], corpus.shape[1], 256)
    x = torch.from_numpy(np.concatenate([O.gather_utterances(corpus, i) for i in (a, p, n)]))[:args.rows]
    allset = set(ALL)
    ref = run(sd, x, set(), set())
    print(f"reference: max |e| {float(ref.abs().max()):.3f}")
    e_all = run(sd, x, allset, allset)
    print(f"fp16 everywhere (filters + stored activations): {err(e_all, ref):.3e}")
    print(f"fp16 filters only:      {err(run(sd, x, allset, set()), ref):.3e}")
    print(f"fp16 activations only:  {err(run(sd, x, set(), allset), ref):.3e}")
    if not args.quick:
        print("layer                      only-this(w)  only-this(a)  only-this(w+a)  all-but-this(w+a)")
        for nm in ALL:
            one = {nm}
            print(f"{nm:26s} {err(run(sd, x, one, set()), ref):.3e}     {err(run(sd, x, set(), one), ref):.3e}     "
                  f"{err(run(sd, x, one, one), ref):.3e}       {err(run(sd, x, allset - one, allset - one), ref):.3e}", flush=True)
        for i in range(1, 5):
            st = {nm for nm in ALL if nm in (f"model.conv{i}", f"model.layer{i}.0.conv1", f"model.layer{i}.0.conv2")}
            print(f"all but stage {i}: {err(run(sd, x, allset - st, allset - st), ref):.3e}   only stage {i}: {err(run(sd, x, st, st), ref):.3e}")
    return sd, x, ref


if __name__ == "__main__":
    main()
