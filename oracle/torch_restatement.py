"""TEST / BASELINE INFRASTRUCTURE -- torch-functional restatement of the reference forward.

The reference's CPU path *is* torch ATen (oneDNN convolutions): `DeepSpeakerModel.forward`
(/root/reference/model.py:185-218) is a chain of nn.Conv2d / nn.BatchNorm2d / nn.Hardtanh /
nn.AdaptiveAvgPool2d / nn.Linear calls.  This file restates that chain with torch.nn.functional
so that bench.py can time "the reference's own CPU forward" on the GPU box's host cores, where
/root/reference does not exist (cpu_baseline.kind = "port").  It is pinned against the recorded
reference outputs by tests/test_oracle_golden.py::test_torch_restatement.  Never imported by the
product package.
"""
import torch
import torch.nn.functional as F


def forward_eval(sd, x, n_stages: int = 4):
    """sd: reference-keyed state_dict of torch tensors; x: [B,1,T,64] float32 CPU tensor."""
    def bn(t, name):
        return F.batch_norm(t, sd[name + ".running_mean"], sd[name + ".running_var"], sd[name + ".weight"],
                            sd[name + ".bias"], False, 0.1, 1e-5)                     # model.py:188 (eval)

    for i in range(1, n_stages + 1):
        x = F.conv2d(x, sd[f"model.conv{i}.weight"], None, 2, 2)                      # model.py:187,192,197,202
        x = F.hardtanh(bn(x, f"model.bn{i}"), 0.0, 20.0)                              # model.py:188-189
        r = x                                                                         # model.py:67
        y = F.conv2d(x, sd[f"model.layer{i}.0.conv1.weight"], None, 1, 1)             # model.py:69
        y = F.hardtanh(bn(y, f"model.layer{i}.0.bn1"), 0.0, 20.0)                     # model.py:70-71
        y = F.conv2d(y, sd[f"model.layer{i}.0.conv2.weight"], None, 1, 1)             # model.py:73
        y = bn(y, f"model.layer{i}.0.bn2")                                            # model.py:74
        x = F.hardtanh(y + r, 0.0, 20.0)                                              # model.py:79-80
    x = F.adaptive_avg_pool2d(x, (1, None))                                           # model.py:207
    x = x.view(x.size(0), -1)                                                         # model.py:208
    x = F.linear(x, sd["model.fc.weight"], sd["model.fc.bias"])                       # model.py:209
    norm = torch.sqrt(torch.sum(x * x, 1) + 1e-10)                                    # model.py:174-177
    return x / norm.view(-1, 1) * 10                                                  # model.py:179,212


# ---------------------------------------------------------------------------------------------------------------
# Training step: forward (train-mode BatchNorm) + triplet loss + autograd, restated on torch ops so that it can be
# evaluated (a) in float64 and (b) with the clipped-ReLU gradient masks of ANOTHER forward substituted.  (b) is
# what pins a reduced-precision backward tightly: hardtanh's gradient is the indicator 0 < x < 20, a value that sits
# within rounding distance of 0 or 20 takes either branch depending on the arithmetic, and one flipped element
# changes the gradient of every upstream layer (the reference's own fp32 and fp64 runs differ by ~1e-2 for that
# reason alone).  With the checked implementation's masks fed in, both sides differentiate the same piecewise-linear
# function and agree to rounding.
# ---------------------------------------------------------------------------------------------------------------
class _ClipFixedMask(torch.autograd.Function):
    """y = clamp(x, 0, 20) (nn.Hardtanh(0, 20), model.py:36-44); backward: g * mask with a supplied mask instead of
    [0 < x < 20]."""

    @staticmethod
    def forward(ctx, x, mask):
        ctx.save_for_backward(mask)
        return x.clamp(0.0, 20.0)

    @staticmethod
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        return g * mask.to(g.dtype), None


def forward_train(params, x, running=None, masks=None, n_stages: int = 4, taps=None):
    """One train-mode forward (model.py:185-218 under model.train()) of x [B,1,T,64].  `params`: reference-keyed
    tensors (leaves with requires_grad for autograd).  `running`: dict of running_mean / running_var tensors updated
    IN PLACE as nn.BatchNorm2d does (momentum 0.1, unbiased variance), or None.  `masks`: None, or a dict
    {"stage{i}.a" | "stage{i}.b" | "stage{i}.c": bool [B,C,H,W]} of clipped-ReLU gradient masks to use instead of
    this forward's own.  `taps` (dict) receives the post-clip activations under the same keys."""
    def bn(t, name):
        rm = running[name + ".running_mean"] if running is not None else None
        rv = running[name + ".running_var"] if running is not None else None
        return F.batch_norm(t, rm, rv, params[name + ".weight"], params[name + ".bias"], True, 0.1, 1e-5)

    def clip(t, key):
        y = F.hardtanh(t, 0.0, 20.0) if masks is None else _ClipFixedMask.apply(t, masks[key])
        if taps is not None:
            taps[key] = y.detach()
        return y

    for i in range(1, n_stages + 1):
        x = F.conv2d(x, params[f"model.conv{i}.weight"], None, 2, 2)
        x = clip(bn(x, f"model.bn{i}"), f"stage{i}.a")
        r = x
        y = F.conv2d(x, params[f"model.layer{i}.0.conv1.weight"], None, 1, 1)
        y = clip(bn(y, f"model.layer{i}.0.bn1"), f"stage{i}.b")
        y = F.conv2d(y, params[f"model.layer{i}.0.conv2.weight"], None, 1, 1)
        y = bn(y, f"model.layer{i}.0.bn2")
        x = clip(y + r, f"stage{i}.c")
    x = F.adaptive_avg_pool2d(x, (1, None))
    x = x.view(x.size(0), -1)
    x = F.linear(x, params["model.fc.weight"], params["model.fc.bias"])
    norm = torch.sqrt(torch.sum(x * x, 1) + 1e-10)
    return x / norm.view(-1, 1) * 10


def triplet_loss(a, p, n, margin):
    """model.py:13-18, 27-33"""
    eps = 1e-4 / a.size(1)
    d_p = torch.pow(torch.sum(torch.pow(torch.abs(a - p), 2), dim=1) + eps, 0.5)
    d_n = torch.pow(torch.sum(torch.pow(torch.abs(a - n), 2), dim=1) + eps, 0.5)
    return torch.mean(torch.clamp(margin + d_p - d_n, min=0.0))


def triplet_train_step(sd, xs, margin=0.1, masks=None, dtype=torch.float64, n_stages: int = 4, ge=None):
    """train_triplet.py:215-223: out_a, out_p, out_n = model(data_a), model(data_p), model(data_n) in train mode (three
    BatchNorm statistic sets, three running-statistics updates), TripletMarginLoss, backward.  `masks`: list of three
    mask dicts (one per member) or None.  `ge`: instead of the loss, backpropagate these three embedding gradients.
    Returns dict(loss, embeddings [3], grads {name: tensor}, running {name: tensor}, acts [3] {key: post-clip})."""
    names = [k for k in sd if ("running" not in k and "num_batches" not in k and "classifier" not in k)]
    params = {k: sd[k].detach().to(dtype).clone().requires_grad_(True) for k in names}
    running = {k: sd[k].detach().to(dtype).clone() for k in sd if "running" in k}
    embs, acts = [], []
    for g, x in enumerate(xs):
        taps = {}
        embs.append(forward_train(params, x.to(dtype), running, None if masks is None else masks[g], n_stages, taps))
        acts.append(taps)
    if ge is None:
        loss = triplet_loss(embs[0], embs[1], embs[2], margin)
        loss.backward()
    else:
        loss = sum((e * g_.to(dtype)).sum() for e, g_ in zip(embs, ge))
        loss.backward()
    return {"loss": loss.detach(), "embeddings": [e.detach() for e in embs],
            "grads": {k: v.grad for k, v in params.items() if v.grad is not None}, "running": running, "acts": acts}
