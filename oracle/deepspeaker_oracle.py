"""CPU oracle for the Deep Speaker embedding path -- TEST INFRASTRUCTURE ONLY.

This file is a numpy restatement of the arithmetic that the reference
(qqueing/DeepSpeaker-pytorch, mounted read-only at /root/reference) executes
for the hot path named in BASELINE.json.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import it; the product package never does.

Pinning: the reference ships no tests or golden vectors,
so the pin is tests/golden/*.npz, produced by tests/golden/make_golden.py which
imports the *unmodified* reference model.py in the build container and records
its outputs on seeded inputs.  tests/test_oracle_golden.py checks every
function below against those fixtures.

Every function cites the reference file:line it restates (paths relative to
/root/reference).  All arrays are numpy; `dtype` selects float32 (what the
reference computes in) or float64 (a tighter checker for fp32 kernels).

Layout at this level is the reference's: activations NCHW [B, C, T, F],
conv weights OIHW, linear weights [out, in].
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

# ----------------------------------------------------------------------------
# constants of the path
# ----------------------------------------------------------------------------
CLIP_MAX = 20.0           # model.py:36-39  ReLU(nn.Hardtanh)(0, 20)
BN_EPS = 1e-5             # torch.nn.BatchNorm2d default, model.py:59,62,94,...
BN_MOMENTUM = 0.1         # torch.nn.BatchNorm2d default
L2_EPS = 1e-10            # model.py:176
ALPHA = 10.0              # model.py:212
STAGE_CHANNELS = (64, 128, 256, 512)   # model.py:93-107


# ----------------------------------------------------------------------------
# deterministic parameter / input generation shared by golden maker, tests,
# bench and smoke (legacy RandomState streams are frozen by numpy policy)
# ----------------------------------------------------------------------------
def make_state_dict(seed: int, num_classes: int = 16, n_stages: int = 4,
                    randomize_bn: bool = True) -> Dict[str, np.ndarray]:
    """Seeded parameters with the reference's shapes and init scale
    (model.py:114-120, 163-167; SURVEY Appendix A key names).  With
    randomize_bn the BN affine/running stats are perturbed so that eval-mode
    normalisation is actually exercised."""
    rs = np.random.RandomState(seed)
    sd: Dict[str, np.ndarray] = {}

    def conv(name, co, ci, k):
        std = math.sqrt(2.0 / (k * k * co))                   # model.py:116-117
        sd[name + ".weight"] = (rs.randn(co, ci, k, k) * std).astype(np.float32)

    def bn(name, c):
        if randomize_bn:
            sd[name + ".weight"] = rs.uniform(0.5, 1.5, c).astype(np.float32)
            sd[name + ".bias"] = (rs.randn(c) * 0.1).astype(np.float32)
            sd[name + ".running_mean"] = (rs.randn(c) * 0.1).astype(np.float32)
            sd[name + ".running_var"] = rs.uniform(0.5, 1.5, c).astype(np.float32)
        else:
            sd[name + ".weight"] = np.ones(c, np.float32)      # model.py:119
            sd[name + ".bias"] = np.zeros(c, np.float32)       # model.py:120
            sd[name + ".running_mean"] = np.zeros(c, np.float32)
            sd[name + ".running_var"] = np.ones(c, np.float32)
        sd[name + ".num_batches_tracked"] = np.zeros((), np.int64)

    cin = 1
    for s in range(n_stages):
        c = STAGE_CHANNELS[s]
        i = s + 1
        conv(f"model.conv{i}", c, cin, 5)
        bn(f"model.bn{i}", c)
        conv(f"model.layer{i}.0.conv1", c, c, 3)
        bn(f"model.layer{i}.0.bn1", c)
        conv(f"model.layer{i}.0.conv2", c, c, 3)
        bn(f"model.layer{i}.0.bn2", c)
        cin = c
    k = 1.0 / math.sqrt(2048.0)                                # nn.Linear default
    sd["model.fc.weight"] = rs.uniform(-k, k, (512, 2048)).astype(np.float32)
    sd["model.fc.bias"] = rs.uniform(-k, k, 512).astype(np.float32)
    k = 1.0 / math.sqrt(512.0)
    sd["model.classifier.weight"] = rs.uniform(-k, k, (num_classes, 512)).astype(np.float32)
    sd["model.classifier.bias"] = rs.uniform(-k, k, num_classes).astype(np.float32)
    return sd


def make_input(seed: int, batch: int, frames: int = 160, scale: float = 1.0) -> np.ndarray:
    """Synthetic filterbank batch [B,1,T,64]."""
    rs = np.random.RandomState(seed)
    return (rs.randn(batch, 1, frames, 64) * scale).astype(np.float32)


def make_speaker_corpus(seed: int, n_speakers: int, utts_per_speaker: int, frames: int = 160,
                        scale: float = 12.0, mix: Tuple[float, float, float] = (0.7, 0.25, 0.65)) -> np.ndarray:
    """Speaker-structured synthetic filterbank features [S, U, frames, 64] with the statistics SURVEY 8(d) gives for
    real inputs: `20*log10` mel energies, per-bin mean-removed, NOT variance-scaled (audio_processing.py:17,29), i.e.
    std ~ 10-20.  Speaker s has a smooth spectral envelope (what a speaker embedding can learn: the temporal mean of
    an utterance recovers it), every utterance a smaller smooth offset of its own (session / channel), every frame
    white noise on top; `mix` = weights of (speaker envelope, utterance offset, frame noise) -- a smaller first weight
    makes the speakers harder to tell apart.  Frozen legacy RandomState stream."""
    rs = np.random.RandomState(seed)

    def smooth(v):
        k = np.array([1, 4, 6, 4, 1], np.float64) / 16.0
        for _ in range(3):
            v = np.apply_along_axis(lambda r: np.convolve(np.pad(r, 2, mode="edge"), k, mode="valid"), -1, v)
        return v / (v.std(axis=-1, keepdims=True) + 1e-12)

    env = smooth(rs.randn(n_speakers, 64))                              # [S, 64]
    off = smooth(rs.randn(n_speakers, utts_per_speaker, 64))            # [S, U, 64]
    noise = rs.randn(n_speakers, utts_per_speaker, frames, 64)
    x = mix[0] * env[:, None, None, :] + mix[1] * off[:, :, None, :] + mix[2] * noise
    return (x * scale).astype(np.float32)


def sample_triplets(seed: int, n_speakers: int, utts_per_speaker: int, n_triplets: int):
    """(anchor, positive, negative) as (speaker, utterance) index pairs + the speaker ids (c1 of anchor / positive, c2 of
    the negative): anchor and positive are two different utterances of one speaker, the negative belongs to another
    speaker -- what DeepSpeakerDataset_dynamic.generate_triplets_call draws (DeepSpeakerDataset_dynamic.py:28-52),
    without its never-the-last-utterance quirk."""
    rs = np.random.RandomState(seed)
    a = np.empty((n_triplets, 2), np.int64)
    p, n = np.empty_like(a), np.empty_like(a)
    for t in range(n_triplets):
        c1 = rs.randint(0, n_speakers)
        c2 = (c1 + 1 + rs.randint(0, n_speakers - 1)) % n_speakers
        ua, up = rs.choice(utts_per_speaker, 2, replace=False)
        a[t], p[t], n[t] = (c1, ua), (c1, up), (c2, rs.randint(0, utts_per_speaker))
    return a, p, n, a[:, 0].copy(), n[:, 0].copy()


def gather_utterances(corpus: np.ndarray, idx: np.ndarray) -> np.ndarray:
    """[S, U, T, 64] corpus, [N, 2] (speaker, utterance) -> the model's input layout [N, 1, T, 64]"""
    return np.ascontiguousarray(corpus[idx[:, 0], idx[:, 1]][:, None])


# ----------------------------------------------------------------------------
# forward primitives
# ----------------------------------------------------------------------------
def conv_out_size(n: int, k: int, stride: int, pad: int) -> int:
    return (n + 2 * pad - k) // stride + 1


def _im2col(x: np.ndarray, k: int, stride: int, pad: int) -> Tuple[np.ndarray, int, int]:
    b, c, h, w = x.shape
    ho, wo = conv_out_size(h, k, stride, pad), conv_out_size(w, k, stride, pad)
    xp = np.pad(x, ((0, 0), (0, 0), (pad, pad), (pad, pad)))
    cols = np.empty((b, c, k, k, ho, wo), x.dtype)
    for i in range(k):
        for j in range(k):
            cols[:, :, i, j] = xp[:, :, i:i + stride * ho:stride, j:j + stride * wo:stride]
    return cols, ho, wo


def conv2d(x: np.ndarray, w: np.ndarray, stride: int, pad: int) -> np.ndarray:
    """nn.Conv2d(bias=False) -- model.py:47-50 (3x3 s1 p1), :93,98,102,106 (5x5 s2 p2)."""
    co, ci, k, _ = w.shape
    cols, ho, wo = _im2col(x, k, stride, pad)
    b = x.shape[0]
    a = cols.transpose(0, 4, 5, 1, 2, 3).reshape(b * ho * wo, ci * k * k)
    y = a @ w.reshape(co, ci * k * k).T
    return np.ascontiguousarray(y.reshape(b, ho, wo, co).transpose(0, 3, 1, 2))


def bn_eval(x, gamma, beta, mean, var):
    """nn.BatchNorm2d in eval mode -- model.py:70,74,188,193,198,203."""
    inv = 1.0 / np.sqrt(var.astype(x.dtype) + x.dtype.type(BN_EPS))
    return (x - mean.astype(x.dtype)[None, :, None, None]) * (inv * gamma.astype(x.dtype))[None, :, None, None] \
        + beta.astype(x.dtype)[None, :, None, None]


def bn_train(x, gamma, beta, running_mean, running_var):
    """nn.BatchNorm2d in train mode: biased batch variance for normalisation,
    unbiased for the running update with momentum 0.1.
    Returns (y, new_running_mean, new_running_var, batch_mean, batch_invstd)."""
    n = x.shape[0] * x.shape[2] * x.shape[3]
    x64 = x.astype(np.float64)
    mean = x64.mean(axis=(0, 2, 3))
    var = x64.var(axis=(0, 2, 3))
    invstd = 1.0 / np.sqrt(var + BN_EPS)
    y = (x64 - mean[None, :, None, None]) * (invstd * gamma)[None, :, None, None] + beta[None, :, None, None]
    unbiased = var * (n / max(n - 1, 1))
    new_rm = (1 - BN_MOMENTUM) * running_mean + BN_MOMENTUM * mean
    new_rv = (1 - BN_MOMENTUM) * running_var + BN_MOMENTUM * unbiased
    return (y.astype(x.dtype), new_rm.astype(np.float32), new_rv.astype(np.float32),
            mean, invstd)


def clip_relu(x):
    """ReLU(nn.Hardtanh)(0,20) -- model.py:36-44."""
    return np.clip(x, 0.0, CLIP_MAX)


def avgpool_time(x):
    """nn.AdaptiveAvgPool2d((1,None)) + view -- model.py:111,207-208.
    [B,C,T',F'] -> [B, C*F'] with index c*F'+f."""
    return x.mean(axis=2).reshape(x.shape[0], -1)


def linear(x, w, b):
    """nn.Linear -- model.py:164,209 (fc), :167,222 (classifier)."""
    return x @ w.T.astype(x.dtype) + b.astype(x.dtype)


def l2_norm_scale(x):
    """DeepSpeakerModel.l2_norm x alpha -- model.py:172-183, 210-213."""
    normp = (x * x).sum(axis=1) + x.dtype.type(L2_EPS)
    return x / np.sqrt(normp)[:, None] * x.dtype.type(ALPHA)


# ----------------------------------------------------------------------------
# whole forward
# ----------------------------------------------------------------------------
def _bn(sd, name, x, train, new_stats, cache):
    g, b = sd[name + ".weight"], sd[name + ".bias"]
    rm, rv = sd[name + ".running_mean"], sd[name + ".running_var"]
    if not train:
        return bn_eval(x, g, b, rm, rv)
    y, nrm, nrv, mean, invstd = bn_train(x, g, b, rm, rv)
    if new_stats is not None:
        new_stats[name + ".running_mean"] = nrm
        new_stats[name + ".running_var"] = nrv
        new_stats[name + ".num_batches_tracked"] = sd[name + ".num_batches_tracked"] + 1
    if cache is not None:
        cache[name] = (x, mean, invstd)
    return y


def forward(sd: Dict[str, np.ndarray], x: np.ndarray, train: bool = False,
            n_stages: int = 4, dtype=np.float32,
            new_stats: Optional[dict] = None, cache: Optional[dict] = None,
            taps: Optional[dict] = None) -> np.ndarray:
    """DeepSpeakerModel.forward -- model.py:185-218 (n_stages=4), or the
    "ResCNN-small" prefix of BASELINE.json configs[0] (n_stages=2: stages 1-2,
    pool, same fc since 128 ch x 16 bins = 2048).

    `new_stats` (dict) receives the updated BN running statistics in train
    mode; `cache` receives what backward() needs; `taps` receives named
    intermediate activations (NCHW) for per-layer kernel checks."""
    x = x.astype(dtype)
    P = {k: (v.astype(dtype) if v.dtype.kind == "f" else v) for k, v in sd.items()}
    for s in range(n_stages):
        i = s + 1
        z = conv2d(x, P[f"model.conv{i}.weight"], 2, 2)                        # :187,192,197,202
        if cache is not None:
            cache[f"model.conv{i}.in"] = x
        x = clip_relu(_bn(P, f"model.bn{i}", z, train, new_stats, cache))      # :188-189
        if taps is not None:
            taps[f"stage{i}.a"] = x
        r = x                                                                  # model.py:67
        if cache is not None:
            cache[f"model.layer{i}.0.conv1.in"] = x
        z = conv2d(x, P[f"model.layer{i}.0.conv1.weight"], 1, 1)               # :69
        y = clip_relu(_bn(P, f"model.layer{i}.0.bn1", z, train, new_stats, cache))   # :70-71
        if taps is not None:
            taps[f"stage{i}.b"] = y
        if cache is not None:
            cache[f"model.layer{i}.0.conv2.in"] = y
        z = conv2d(y, P[f"model.layer{i}.0.conv2.weight"], 1, 1)               # :73
        y = _bn(P, f"model.layer{i}.0.bn2", z, train, new_stats, cache)        # :74
        x = clip_relu(y + r)                                                   # :79-80
        if taps is not None:
            taps[f"stage{i}.c"] = x
    if cache is not None:
        cache["pool.in"] = x
    p = avgpool_time(x)                                                        # :207-208
    f = linear(p, P["model.fc.weight"], P["model.fc.bias"])                    # :209
    if cache is not None:
        cache["fc.in"] = p
        cache["fc.out"] = f
    e = l2_norm_scale(f)                                                       # :210-213
    if taps is not None:
        taps["pooled"] = p
        taps["fc"] = f
    return e


def forward_classifier(sd, x, **kw):
    """DeepSpeakerModel.forward_classifier -- model.py:220-223."""
    e = forward(sd, x, **kw)
    return linear(e, sd["model.classifier.weight"].astype(e.dtype),
                  sd["model.classifier.bias"].astype(e.dtype))


# ----------------------------------------------------------------------------
# loss side: pairwise distance, triplet margin loss, triplet filter, CE
# ----------------------------------------------------------------------------
def pairwise_distance(x1: np.ndarray, x2: np.ndarray, p: int = 2) -> np.ndarray:
    """PairwiseDistance.forward -- model.py:13-18.  eps = 1e-4 / D inside the root."""
    assert x1.shape == x2.shape                                  # model.py:14
    eps = x1.dtype.type(1e-4 / x1.shape[1])
    diff = np.abs(x1 - x2)
    out = np.power(diff, p).sum(axis=1)
    return np.power(out + eps, x1.dtype.type(1.0 / p))


def triplet_margin_loss(a, p, n, margin: float):
    """TripletMarginLoss.forward -- model.py:27-33."""
    d_p = pairwise_distance(a, p)
    d_n = pairwise_distance(a, n)
    hinge = np.maximum(a.dtype.type(margin) + d_p - d_n, a.dtype.type(0.0))
    return hinge.mean(dtype=np.float64).astype(a.dtype), d_p, d_n


def triplet_filter(d_p: np.ndarray, d_n: np.ndarray, margin: float):
    """The reference's "mining": train_triplet.py:251-262.
    Returns (selected indices ascending, n_correct, mean(d_n - d_p))."""
    diff = (d_n - d_p).astype(np.float32)
    mask = diff < np.float32(margin)                             # :253 strict <
    idx = np.where(mask)[0]                                      # :262 ascending
    n_correct = int((~mask).sum())                               # :256-257
    return idx.astype(np.int64), n_correct, float(diff.mean(dtype=np.float64))  # :259-260


def cross_entropy(logits: np.ndarray, labels: np.ndarray) -> np.ndarray:
    """nn.CrossEntropyLoss (mean) -- train_triplet.py:281-285."""
    z = logits.astype(np.float64)
    z = z - z.max(axis=1, keepdims=True)
    lse = np.log(np.exp(z).sum(axis=1))
    nll = lse - z[np.arange(len(labels)), labels]
    return nll.mean().astype(logits.dtype)


def test_scores(emb_a: np.ndarray, emb_p: np.ndarray, crops: int) -> np.ndarray:
    """test(): mean over `crops` crop-pair distances -- train_triplet.py:347-350."""
    d = pairwise_distance(emb_a, emb_p)
    return d.reshape(-1, crops).mean(axis=1)


def roc_sweep(distances: np.ndarray, labels: np.ndarray, thresholds: np.ndarray):
    """eval_metrics.py:16-50: per threshold tp/fp with predict_issame = dist < threshold; best threshold =
    first argmax of accuracy.  Returns (tp[], fp[], best_index, tpr, fpr, acc)."""
    issame = labels.astype(bool)
    d32 = distances.astype(np.float32)
    tp = np.array([np.sum(np.logical_and(np.less(d32, np.float32(t)), issame)) for t in thresholds])
    fp = np.array([np.sum(np.logical_and(np.less(d32, np.float32(t)), ~issame)) for t in thresholds])
    n_same, n_diff = issame.sum(), (~issame).sum()
    acc = (tp + (n_diff - fp)) / float(d32.size)                       # eval_metrics.py:49
    best = int(np.argmax(acc))                                          # eval_metrics.py:33
    tpr = 0 if n_same == 0 else tp[best] / float(n_same)               # eval_metrics.py:47
    fpr = 0 if n_diff == 0 else fp[best] / float(n_diff)               # eval_metrics.py:48
    return tp, fp, best, float(tpr), float(fpr), float(acc[best])


def equal_error_rate(tp: np.ndarray, fp: np.ndarray, n_same: int, n_diff: int) -> float:
    """EER from a threshold sweep (NEW: the reference computes none): first threshold where
    FPR >= FNR, linearly interpolated with the previous one."""
    fpr = fp / max(n_diff, 1)
    fnr = 1.0 - tp / max(n_same, 1)
    i = int(np.argmax(fpr >= fnr))
    if i == 0:
        return float(0.5 * (fpr[0] + fnr[0]))
    d0, d1 = fnr[i - 1] - fpr[i - 1], fpr[i] - fnr[i]
    w = d0 / (d0 + d1) if (d0 + d1) > 0 else 0.0
    return float(fpr[i - 1] + w * (fpr[i] - fpr[i - 1]))


def mine_semihard(anchor: np.ndarray, d_p: np.ndarray, anchor_label: np.ndarray,
                  cand: np.ndarray, cand_label: np.ndarray) -> np.ndarray:
    """Cross-GPU semi-hard negative search (NEW capability, no reference
    counterpart -- SURVEY F4, 8(e)).  For anchor i choose the candidate j with
    cand_label[j] != anchor_label[i] and d(a_i, x_j) > d_p[i] that minimises
    d(a_i, x_j); if none is semi-hard, the closest different-speaker candidate;
    ties -> lowest j; -1 if no different-speaker candidate exists.
    Distances use the reference's pairwise_distance arithmetic (eps inside
    the root, fp32)."""
    n, m = anchor.shape[0], cand.shape[0]
    out = np.full(n, -1, np.int64)
    eps = np.float32(1e-4 / anchor.shape[1])
    for i in range(n):
        diff = (anchor[i][None, :] - cand).astype(np.float32)
        d = np.sqrt((diff * diff).sum(axis=1, dtype=np.float32) + eps)
        ok = cand_label != anchor_label[i]
        if not ok.any():
            continue
        semi = ok & (d > d_p[i])
        pool = semi if semi.any() else ok
        dd = np.where(pool, d, np.float32(np.inf))
        out[i] = int(np.argmin(dd))
    return out


# ----------------------------------------------------------------------------
# backward (restates what torch autograd does for loss.backward(),
# train_triplet.py:223,290 -- SURVEY 8(a) a13)
# ----------------------------------------------------------------------------
def pairwise_distance_bwd(x1, x2, d, gd):
    """d = sqrt(sum((x1-x2)^2)+eps);  dd/dx1 = (x1-x2)/d."""
    g = (gd / d)[:, None] * (x1 - x2)
    return g, -g


def triplet_margin_loss_bwd(a, p, n, margin):
    """Gradient of mean(clamp(margin + d_p - d_n, min=0)); subgradient at 0 is 1
    (torch.clamp(min=0) passes gradient where input >= 0 -- SURVEY a10)."""
    d_p = pairwise_distance(a, p)
    d_n = pairwise_distance(a, n)
    act = ((a.dtype.type(margin) + d_p - d_n) >= 0).astype(a.dtype) / a.shape[0]
    ga1, gp = pairwise_distance_bwd(a, p, d_p, act)
    ga2, gn = pairwise_distance_bwd(a, n, d_n, -act)
    return ga1 + ga2, gp, gn


def l2_norm_scale_bwd(f, ge):
    """e = alpha * f / sqrt(sum f^2 + eps)."""
    nrm = np.sqrt((f * f).sum(axis=1) + f.dtype.type(L2_EPS))[:, None]
    dot = (ge * f).sum(axis=1)[:, None]
    return f.dtype.type(ALPHA) * (ge / nrm - f * dot / nrm ** 3)


def conv2d_bwd(x, w, gy, stride, pad, need_gx=True):
    co, ci, k, _ = w.shape
    b, _, h, wd = x.shape
    cols, ho, wo = _im2col(x, k, stride, pad)
    a = cols.transpose(0, 4, 5, 1, 2, 3).reshape(b * ho * wo, ci * k * k)
    g = gy.transpose(0, 2, 3, 1).reshape(b * ho * wo, co)
    gw = (g.T @ a).reshape(co, ci, k, k)
    gx = None
    if need_gx:
        ga = (g @ w.reshape(co, -1)).reshape(b, ho, wo, ci, k, k).transpose(0, 3, 4, 5, 1, 2)
        gxp = np.zeros((b, ci, h + 2 * pad, wd + 2 * pad), x.dtype)
        for i in range(k):
            for j in range(k):
                gxp[:, :, i:i + stride * ho:stride, j:j + stride * wo:stride] += ga[:, :, i, j]
        gx = gxp[:, :, pad:pad + h, pad:pad + wd]
    return gx, gw


def bn_train_bwd(x, mean, invstd, gamma, gy):
    n = x.shape[0] * x.shape[2] * x.shape[3]
    xh = (x - mean[None, :, None, None]) * invstd[None, :, None, None]
    gb = gy.sum(axis=(0, 2, 3))
    gg = (gy * xh).sum(axis=(0, 2, 3))
    gx = (gamma * invstd)[None, :, None, None] * (
        gy - gb[None, :, None, None] / n - xh * gg[None, :, None, None] / n)
    return gx, gg, gb


def clip_bwd(out, g):
    """Hardtanh(0,20) backward: passes where 0 < x < 20 (strict).
    `out` is the clipped output, equivalent as a mask."""
    return g * ((out > 0) & (out < CLIP_MAX))


def backward(sd, cache, x_in, emb_grad, n_stages: int = 4, dtype=np.float64) -> Dict[str, np.ndarray]:
    """Backward of forward(train=True) given dL/d(embedding).  Requires the
    `cache` filled by forward(..., train=True, cache=cache).  Returns parameter
    gradients keyed like the state_dict."""
    P = {k: (v.astype(dtype) if v.dtype.kind == "f" else v) for k, v in sd.items()}
    grads: Dict[str, np.ndarray] = {}
    ge = emb_grad.astype(dtype)
    f = cache["fc.out"].astype(dtype)
    gf = l2_norm_scale_bwd(f, ge)
    p = cache["fc.in"].astype(dtype)
    grads["model.fc.weight"] = gf.T @ p
    grads["model.fc.bias"] = gf.sum(axis=0)
    gp = gf @ P["model.fc.weight"]
    xl = cache["pool.in"].astype(dtype)
    b, c, t, fr = xl.shape
    g = np.broadcast_to(gp.reshape(b, c, 1, fr) / t, xl.shape).copy()
    out = xl
    for s in reversed(range(n_stages)):
        i = s + 1
        # out = clip(bn2(conv2(y)) + r)
        g = clip_bwd(out, g)
        g_res = g
        name = f"model.layer{i}.0.bn2"
        z, mean, invstd = cache[name]
        gz, gg, gb = bn_train_bwd(z.astype(dtype), mean, invstd, P[name + ".weight"], g)
        grads[name + ".weight"], grads[name + ".bias"] = gg, gb
        y = cache[f"model.layer{i}.0.conv2.in"].astype(dtype)
        gy, gw = conv2d_bwd(y, P[f"model.layer{i}.0.conv2.weight"], gz, 1, 1)
        grads[f"model.layer{i}.0.conv2.weight"] = gw
        gy = clip_bwd(y, gy)
        name = f"model.layer{i}.0.bn1"
        z, mean, invstd = cache[name]
        gz, gg, gb = bn_train_bwd(z.astype(dtype), mean, invstd, P[name + ".weight"], gy)
        grads[name + ".weight"], grads[name + ".bias"] = gg, gb
        r = cache[f"model.layer{i}.0.conv1.in"].astype(dtype)
        gr, gw = conv2d_bwd(r, P[f"model.layer{i}.0.conv1.weight"], gz, 1, 1)
        grads[f"model.layer{i}.0.conv1.weight"] = gw
        g = gr + g_res
        # r = clip(bn_i(conv_i(x)))
        g = clip_bwd(r, g)
        name = f"model.bn{i}"
        z, mean, invstd = cache[name]
        gz, gg, gb = bn_train_bwd(z.astype(dtype), mean, invstd, P[name + ".weight"], g)
        grads[name + ".weight"], grads[name + ".bias"] = gg, gb
        xin = cache[f"model.conv{i}.in"].astype(dtype)
        g, gw = conv2d_bwd(xin, P[f"model.conv{i}.weight"], gz, 2, 2, need_gx=(s > 0))
        grads[f"model.conv{i}.weight"] = gw
        out = xin
    return grads
