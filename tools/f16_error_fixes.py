"""Candidate fixes for the fp16 eval path's embedding error on trained networks, evaluated with the CPU simulation of
tools/f16_error_budget.py (same trained fixture, same measure).  python tools/f16_error_fixes.py [--opt sgd]"""
import argparse
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import f16_error_budget as EB      # noqa: E402
from f16_error_budget import ALL, O, err, h   # noqa: E402


def forward(sd, x, wq, round_conv_in=True, round_residual=True, round_final_residual=True):
    """wq: {layer: filter}; activations: conv inputs rounded to fp16 if round_conv_in; the residual operand rounded if
    round_residual (the stored tensor is ONE tensor in the product: both flags on = today's path)."""
    def bn(t, name):
        return F.batch_norm(t, sd[name + ".running_mean"], sd[name + ".running_var"], sd[name + ".weight"],
                            sd[name + ".bias"], False, 0.1, 1e-5)
    rc = (lambda t: h(t)) if round_conv_in else (lambda t: t)
    for i in range(1, 5):
        if i == 1:
            x = F.conv2d(x, sd["model.conv1.weight"], None, 2, 2)
        else:
            x = F.conv2d(rc(x), wq[f"model.conv{i}"], None, 2, 2)
        x = F.hardtanh(bn(x, f"model.bn{i}"), 0.0, 20.0)
        r = h(x) if round_residual else x
        y = F.conv2d(rc(x), wq[f"model.layer{i}.0.conv1"], None, 1, 1)
        y = F.hardtanh(bn(y, f"model.layer{i}.0.bn1"), 0.0, 20.0)
        y = F.conv2d(rc(y), wq[f"model.layer{i}.0.conv2"], None, 1, 1)
        y = bn(y, f"model.layer{i}.0.bn2")
        x = F.hardtanh(y + r, 0.0, 20.0)
    x = F.adaptive_avg_pool2d(x, (1, None)).reshape(x.size(0), -1)
    x = F.linear(x, sd["model.fc.weight"], sd["model.fc.bias"])
    norm = torch.sqrt(torch.sum(x * x, 1) + 1e-10)
    return x / norm.view(-1, 1) * 10


def run(sd, x, wq, **kw):
    with torch.no_grad():
        return torch.cat([forward(sd, x[i:i + 64], wq, **kw) for i in range(0, x.shape[0], 64)])


def channel_means(sd, x):
    """mean input activation per channel of every stage convolution (f32 forward over a calibration batch)"""
    means = {}

    def bn(t, name):
        return F.batch_norm(t, sd[name + ".running_mean"], sd[name + ".running_var"], sd[name + ".weight"],
                            sd[name + ".bias"], False, 0.1, 1e-5)
    with torch.no_grad():
        for i in range(1, 5):
            if i > 1:
                means[f"model.conv{i}"] = x.mean(dim=(0, 2, 3))
            x = F.conv2d(x, sd[f"model.conv{i}.weight"], None, 2, 2)
            x = F.hardtanh(bn(x, f"model.bn{i}"), 0.0, 20.0)
            r = x
            means[f"model.layer{i}.0.conv1"] = x.mean(dim=(0, 2, 3))
            y = F.conv2d(x, sd[f"model.layer{i}.0.conv1.weight"], None, 1, 1)
            y = F.hardtanh(bn(y, f"model.layer{i}.0.bn1"), 0.0, 20.0)
            means[f"model.layer{i}.0.conv2"] = y.mean(dim=(0, 2, 3))
            y = F.conv2d(y, sd[f"model.layer{i}.0.conv2.weight"], None, 1, 1)
            x = F.hardtanh(bn(y, f"model.layer{i}.0.bn2") + r, 0.0, 20.0)
    return means


def diffuse(w, m):
    """Error-diffusion rounding of a filter bank [O, I, kh, kw] to fp16: per output channel, walk the contraction axis
    (taps outer, input channels inner) and round each weight so that the running sum of (rounded - exact) * m[channel]
    stays within half an ulp-weight of zero: the response to the MEAN input is then exact to one rounding instead of
    sqrt(K) of them.  Every rounded weight is one of the two fp16 neighbours of the exact one."""
    O_, I, kh, kw = w.shape
    wk = w.permute(0, 2, 3, 1).reshape(O_, kh * kw * I).double()            # taps outer, channels inner
    mk = m.double().repeat(kh * kw).clamp_min(1e-6)
    lo = wk.float().half()                                                   # nearest
    # the two neighbours: nearest, and the next one on the other side of the exact value
    near = lo.double()
    other = torch.nextafter(lo, torch.where(wk > near, torch.tensor(float("inf")).half(), torch.tensor(float("-inf")).half())).double()
    out = torch.empty_like(wk)
    acc = torch.zeros(O_, dtype=torch.float64)
    for k in range(wk.shape[1]):
        e_near = (near[:, k] - wk[:, k]) * mk[k]
        e_oth = (other[:, k] - wk[:, k]) * mk[k]
        pick_oth = (acc + e_oth).abs() < (acc + e_near).abs()
        out[:, k] = torch.where(pick_oth, other[:, k], near[:, k])
        acc = acc + torch.where(pick_oth, e_oth, e_near)
    return out.float().reshape(O_, kh, kw, I).permute(0, 3, 1, 2).contiguous(), float(acc.abs().max())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--opt", default="sgd")
    ap.add_argument("--rows", type=int, default=768)
    args = ap.parse_args()
    torch.set_num_threads(os.cpu_count() or 8)
    sd, corpus = torch.load(f"/tmp/f16_budget_{args.opt}.pt", weights_only=False)
    a, p, n, _, _ = O.sample_triplets(9000, corpus.shape[0], corpus.shape[1], 256)
    x = torch.from_numpy(np.concatenate([O.gather_utterances(corpus, i) for i in (a, p, n)]))[:args.rows]
    w32 = {nm: sd[nm + ".weight"] for nm in ALL}
    w16 = {nm: h(sd[nm + ".weight"]) for nm in ALL}
    ref = run(sd, x, w32, round_conv_in=False, round_residual=False)
    print(f"today (fp16 filters, fp16 stored activations): {err(run(sd, x, w16), ref):.3e}")
    print(f"activations only:                              {err(run(sd, x, w32), ref):.3e}")
    print(f"  ... conv inputs rounded, residual exact:     {err(run(sd, x, w32, round_residual=False), ref):.3e}")
    print(f"  ... residual rounded, conv inputs exact:     {err(run(sd, x, w32, round_conv_in=False), ref):.3e}")
    print(f"filters only:                                  {err(run(sd, x, w16, round_conv_in=False, round_residual=False), ref):.3e}")
    # error-diffused filters, channel means from a DIFFERENT calibration batch (other triplets of the corpus)
    a2, p2, n2, _, _ = O.sample_triplets(77, corpus.shape[0], corpus.shape[1], 32)
    xc = torch.from_numpy(np.concatenate([O.gather_utterances(corpus, i) for i in (a2, p2, n2)]))
    means = channel_means(sd, xc)
    wd = {}
    for nm in ALL:
        wd[nm], resid = diffuse(sd[nm + ".weight"], means[nm])
        assert float((wd[nm] - w32[nm]).abs().max()) <= 2 * float((w16[nm] - w32[nm]).abs().max()) + 1e-12
    print(f"filters only, error-diffused:                  {err(run(sd, x, wd, round_conv_in=False, round_residual=False), ref):.3e}")
    print(f"error-diffused filters + fp16 activations:     {err(run(sd, x, wd), ref):.3e}")
    print(f"error-diffused filters + exact residual:       {err(run(sd, x, wd, round_residual=False), ref):.3e}")
    print(f"fp16 filters + exact residual:                 {err(run(sd, x, w16, round_residual=False), ref):.3e}")


if __name__ == "__main__":
    main()
