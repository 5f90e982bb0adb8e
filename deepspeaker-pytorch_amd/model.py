"""Drop-in replacement for the reference's `model.py` surface.

Same names, constructor arguments, attributes and `state_dict` keys as
qqueing/DeepSpeaker-pytorch `model.py` (file:line cited per symbol), so that
`train_triplet.py`-style code (`model(data_a)`, `TripletMarginLoss(m).forward(a, p, n)`,
`l2_dist.forward(a, b)`, `model.state_dict()`, `optim.Adagrad(model.parameters())`) runs unchanged.
The arithmetic is done by HIP kernels through `engine.Engine`; the `nn.Conv2d` / `nn.BatchNorm2d` /
`nn.Linear` children below are parameter containers only (they give identical keys, shapes and
initialisation) and are never called.

There is no CPU path: tensors must live on a ROCm device, otherwise a RuntimeError is raised.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn as nn

from . import _native
from .engine import ALPHA, BNParams, Engine, L2_EPS, PRECISIONS, STAGE_CHANNELS
from .precision_guard import F16_GUARD_THRESHOLD, F16Guard

_engine: Optional[Engine] = None


def get_engine() -> Engine:
    """Engine bound to libdeepspeaker_hip.so; raises if the library has not been built."""
    global _engine
    if _engine is None:
        _engine = Engine(_native.load())
    return _engine


def _require_cuda(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise RuntimeError(f"{what}: tensor is on {t.device}; deepspeaker-pytorch_amd computes only on an "
                           "MI355X (ROCm) device and has no CPU fallback -- move the model and inputs with .cuda()")


# ---------------------------------------------------------------------------------------------
# loss side (reference model.py:8-33)
# ---------------------------------------------------------------------------------------------
class _PairwiseDistanceFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x1, x2):
        eng = get_engine()
        x1c, x2c = x1.contiguous(), x2.contiguous()
        d = eng.pairwise_distance(x1c, x2c)
        ctx.save_for_backward(x1c, x2c, d)
        return d

    @staticmethod
    def backward(ctx, gd):
        x1, x2, d = ctx.saved_tensors
        eng = get_engine()
        g1, g2 = torch.empty_like(x1), torch.empty_like(x2)
        gd = gd.contiguous()
        eng.lib.call("ds_pairwise_distance_bwd_f32", eng._p(x1), eng._p(x2), eng._p(d), eng._p(gd), eng._p(g1),
                     eng._p(g2), x1.shape[0], x1.shape[1], eng._stream(x1))
        return g1, g2


class _PairwiseDistancePFn(torch.autograd.Function):
    """any norm p > 0 (the p = 2 of the reference's call sites has its own kernels above)"""

    @staticmethod
    def forward(ctx, x1, x2, p):
        eng = get_engine()
        x1c, x2c = x1.contiguous().float(), x2.contiguous().float()
        d = torch.empty(x1c.shape[0], dtype=torch.float32, device=x1c.device)
        eng.lib.call("ds_pairwise_distance_p_f32", eng._p(x1c), eng._p(x2c), eng._p(d), x1c.shape[0], x1c.shape[1],
                     float(p), eng._stream(x1c))
        ctx.save_for_backward(x1c, x2c, d)
        ctx.p = float(p)
        return d

    @staticmethod
    def backward(ctx, gd):
        x1, x2, d = ctx.saved_tensors
        eng = get_engine()
        g1, g2 = torch.empty_like(x1), torch.empty_like(x2)
        gd = gd.contiguous().float()
        eng.lib.call("ds_pairwise_distance_p_bwd_f32", eng._p(x1), eng._p(x2), eng._p(d), eng._p(gd), eng._p(g1),
                     eng._p(g2), x1.shape[0], x1.shape[1], ctx.p, eng._stream(x1))
        return g1, g2, None


class PairwiseDistance:
    """reference model.py:8-18.  `PairwiseDistance(p).forward(x1, x2)` -> [N] distances
    pow(sum |x1-x2|^p + 1e-4/D, 1/p).  The reference's call sites pass 2 (train_triplet.py:119, model.py:24): that is
    the fused path; any other p > 0 runs the general kernel pair."""

    def __init__(self, p):
        if not float(p) > 0:
            raise ValueError("the norm p must be positive")
        self.norm = p

    def forward(self, x1, x2):
        assert x1.size() == x2.size()                       # reference model.py:14
        _require_cuda(x1, "PairwiseDistance")
        if self.norm != 2:
            return _PairwiseDistancePFn.apply(x1, x2, self.norm)
        return _PairwiseDistanceFn.apply(x1, x2)

    __call__ = forward


class _TripletMarginFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, p, n, margin):
        eng = get_engine()
        a, p, n = a.contiguous(), p.contiguous(), n.contiguous()
        loss, d_p, d_n = eng.triplet_margin(a, p, n, margin)
        ctx.save_for_backward(a, p, n, d_p, d_n)
        ctx.margin = margin
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gl):
        a, p, n, d_p, d_n = ctx.saved_tensors
        eng = get_engine()
        ga, gp, gn = torch.empty_like(a), torch.empty_like(p), torch.empty_like(n)
        gl = gl.reshape(1).contiguous().float()
        eng.lib.call("ds_triplet_margin_bwd_f32", eng._p(a), eng._p(p), eng._p(n), eng._p(d_p), eng._p(d_n),
                     float(ctx.margin), eng._p(gl), eng._p(ga), eng._p(gp), eng._p(gn), a.shape[0], a.shape[1],
                     eng._stream(a))
        return ga, gp, gn, None


class TripletMarginLoss:
    """reference model.py:19-33: mean(clamp(margin + d(a,p) - d(a,n), min=0))."""

    def __init__(self, margin):
        self.margin = margin
        self.pdist = PairwiseDistance(2)

    def forward(self, anchor, positive, negative):
        _require_cuda(anchor, "TripletMarginLoss")
        return _TripletMarginFn.apply(anchor, positive, negative, float(self.margin))

    __call__ = forward


# ---------------------------------------------------------------------------------------------
# softmax pre-training head: classifier GEMM + cross-entropy (model.py:167,220-223; train_triplet.py:277-287)
# ---------------------------------------------------------------------------------------------
def _pad128(n: int) -> int:
    return (n + 127) // 128 * 128


class HeadPack:
    """Kernel-layout copies of the classifier (model.py:167): rows zero-padded to a multiple of 128, packed for the
    forward GEMM and, transposed, for the data gradient.  Built once per parameter version (DeepSpeakerModel._head)."""

    def __init__(self, eng, weight, bias):
        n, k = weight.shape
        self.n, self.k, self.npad = n, k, _pad128(n)
        dev = weight.device
        self.wpad = torch.zeros((self.npad, k), dtype=torch.float32, device=dev)
        self.wpad[:n].copy_(weight.detach())
        self.bpad = torch.zeros(self.npad, dtype=torch.float32, device=dev)
        self.bpad[:n].copy_(bias.detach())
        st = eng._stream(weight)
        self.wp = torch.empty(self.npad * k, dtype=torch.float32, device=dev)
        eng.lib.call("ds_pack_fc_weight_f32", eng._p(self.wpad), eng._p(self.wp), self.npad, k, 1, st)
        self.wd = torch.empty(self.npad * k, dtype=torch.float32, device=dev)
        eng.lib.call("ds_pack_fc_weight_dgrad_f32", eng._p(self.wpad), eng._p(self.wd), self.npad, k, 1, st)


def _head_backward(eng, pack: HeadPack, x, gp):
    """(dx, dW, db) of logits = x W^T + b given the padded logits gradient gp [M, Np]"""
    from ._native import ConvShape
    import ctypes
    m, k = x.shape
    st = eng._stream(x)
    ws = torch.empty(eng.lib.raw("ds_fc_workspace_floats")(m, pack.npad, k), dtype=torch.float32, device=x.device)
    gx = torch.empty((m, k), dtype=torch.float32, device=x.device)
    eng.lib.call("ds_fc_l2norm_fwd_f32", eng._p(gp), eng._p(pack.wd), None, eng._p(ws), eng._p(gx), None, m, pack.npad, k,
                 1.0, 0.0, st)                                      # dx = g Wpad
    shp = ConvShape(1, m, 1, k, pack.npad, 1, 1)                    # dW = g^T x: a 1x1 "convolution" over the M rows
    ws2 = torch.empty(eng.lib.raw("ds_conv_wgrad_workspace_floats")(ctypes.byref(shp)), dtype=torch.float32,
                      device=x.device)
    gw = torch.empty((pack.npad, k), dtype=torch.float32, device=x.device)
    eng.lib.call("ds_conv_wgrad_f32", ctypes.byref(shp), eng._p(x), eng._p(gp), eng._p(ws2), eng._p(gw), 0, st)
    gb = torch.empty(pack.npad, dtype=torch.float32, device=x.device)
    eng.lib.call("ds_colsum_f32", eng._p(gp), eng._p(gb), m, pack.npad, st)
    return gx, gw[:pack.n], gb[:pack.n]


class _LinearHeadFn(torch.autograd.Function):
    """y = x W^T + b on the f32 matrix cores (split-K GEMM of fc_mfma_f32.hip); the returned logits are the
    [M, n_cls] view of the padded [M, Np] buffer."""

    @staticmethod
    def forward(ctx, x, weight, bias, eng, pack):
        x = x.contiguous()
        m, k = x.shape
        ws = torch.empty(eng.lib.raw("ds_fc_workspace_floats")(m, k, pack.npad), dtype=torch.float32, device=x.device)
        out = torch.empty((m, pack.npad), dtype=torch.float32, device=x.device)
        eng.lib.call("ds_fc_l2norm_fwd_f32", eng._p(x), eng._p(pack.wp), eng._p(pack.bpad), eng._p(ws), eng._p(out), None,
                     m, k, pack.npad, 1.0, 0.0, eng._stream(x))
        ctx.save_for_backward(x)
        ctx.eng, ctx.pack = eng, pack
        return out[:, :pack.n]

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        pack = ctx.pack
        gp = torch.zeros((x.shape[0], pack.npad), dtype=torch.float32, device=x.device)
        gp[:, :pack.n].copy_(g)
        gx, gw, gb = _head_backward(ctx.eng, pack, x, gp)
        return gx, gw, gb, None, None


class _HeadCrossEntropyFn(torch.autograd.Function):
    """loss = CrossEntropy(x W^T + b, labels) (model.py:220-223 + train_triplet.py:281-285) with the row maximum,
    log-sum-exp and per-row loss taken in the GEMM's reduction epilogue (ds_fc_ce_fwd_f32)."""

    @staticmethod
    def forward(ctx, x, weight, bias, labels, eng, pack):
        x = x.contiguous()
        m, k = x.shape
        labels = labels.to(torch.int64).contiguous()
        dev = x.device
        ws = torch.empty(eng.lib.raw("ds_fc_workspace_floats")(m, k, pack.npad), dtype=torch.float32, device=dev)
        logits = torch.empty((m, pack.npad), dtype=torch.float32, device=dev)
        row_loss, lse = torch.empty(m, dtype=torch.float32, device=dev), torch.empty(m, dtype=torch.float32, device=dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        eng.lib.call("ds_fc_ce_fwd_f32", eng._p(x), eng._p(pack.wp), eng._p(pack.bpad), eng._p(ws), eng._p(logits),
                     eng._p(labels), eng._p(row_loss), eng._p(lse), eng._p(loss), m, k, pack.npad, pack.n, eng._stream(x))
        ctx.save_for_backward(x, logits, labels, lse)
        ctx.eng, ctx.pack = eng, pack
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gl):
        x, logits, labels, lse = ctx.saved_tensors
        eng, pack = ctx.eng, ctx.pack
        m = x.shape[0]
        gp = torch.zeros((m, pack.npad), dtype=torch.float32, device=x.device)      # pad columns stay zero
        gl = gl.reshape(1).contiguous().float()
        eng.lib.call("ds_cross_entropy_bwd_f32", eng._p(logits), eng._p(labels), eng._p(lse), eng._p(gl), eng._p(gp), m,
                     pack.n, pack.npad, pack.npad, eng._stream(x))
        gx, gw, gb = _head_backward(eng, pack, x, gp)
        return gx, gw, gb, None, None, None


class _CrossEntropyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, eng):
        if logits.stride(1) != 1:
            logits = logits.contiguous()
        m, n = logits.shape
        ld = logits.stride(0)
        labels = labels.to(torch.int64).contiguous()
        row_loss = torch.empty(m, dtype=torch.float32, device=logits.device)
        lse = torch.empty_like(row_loss)
        loss = torch.empty(1, dtype=torch.float32, device=logits.device)
        eng.lib.call("ds_cross_entropy_fwd_f32", eng._p(logits), eng._p(labels), eng._p(row_loss), eng._p(lse),
                     eng._p(loss), m, n, ld, eng._stream(logits))
        ctx.save_for_backward(logits, labels, lse)
        ctx.eng = eng
        return loss.reshape(())

    @staticmethod
    def backward(ctx, gl):
        logits, labels, lse = ctx.saved_tensors
        eng = ctx.eng
        m, n = logits.shape
        d = torch.empty((m, n), dtype=torch.float32, device=logits.device)
        gl = gl.reshape(1).contiguous().float()
        eng.lib.call("ds_cross_entropy_bwd_f32", eng._p(logits), eng._p(labels), eng._p(lse), eng._p(gl), eng._p(d), m,
                     n, logits.stride(0), n, eng._stream(logits))
        return d, None, None


class CrossEntropyLoss:
    """Drop-in for the `nn.CrossEntropyLoss()` of train_triplet.py:281 (mean reduction) on HIP kernels."""

    def forward(self, logits, labels):
        _require_cuda(logits, "CrossEntropyLoss")
        return _CrossEntropyFn.apply(logits, labels, get_engine())

    __call__ = forward


# ---------------------------------------------------------------------------------------------
# parameter containers mirroring the reference's module tree (model.py:36-130)
# ---------------------------------------------------------------------------------------------
class ReLU(nn.Hardtanh):
    """reference model.py:36-44: clipped ReLU, min(max(x,0),20).  Container only; the clip is fused into the
    convolution / normalisation kernels."""

    def __init__(self, inplace=False):
        super().__init__(0, 20, inplace)


def conv3x3(in_planes, out_planes, stride=1):
    """reference model.py:47-50"""
    return nn.Conv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=1, bias=False)


class BasicBlock(nn.Module):
    """reference model.py:53-82 (parameters only)."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = conv3x3(inplanes, planes, stride)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = ReLU(inplace=True)
        self.conv2 = conv3x3(planes, planes)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x):
        raise RuntimeError("BasicBlock is a parameter container; call DeepSpeakerModel.forward")


class myResNet(nn.Module):
    """reference model.py:85-120 (parameters + initialisation only).  `n_stages` < 4 builds the
    "ResCNN-small" prefix of BASELINE.json configs[0]."""

    def __init__(self, block=BasicBlock, layers=(1, 1, 1, 1), n_stages: int = 4):
        super().__init__()
        self.relu = ReLU(inplace=True)
        cin = 1
        for s in range(n_stages):
            c, i = STAGE_CHANNELS[s], s + 1
            setattr(self, f"conv{i}", nn.Conv2d(cin, c, kernel_size=5, stride=2, padding=2, bias=False))
            setattr(self, f"bn{i}", nn.BatchNorm2d(c))
            setattr(self, f"layer{i}", nn.Sequential(block(c, c)))
            cin = c
        self.avgpool = nn.AdaptiveAvgPool2d((1, None))
        for m in self.modules():                                    # reference model.py:114-120
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2. / n))
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def forward(self, x):
        raise RuntimeError("myResNet is a parameter container; call DeepSpeakerModel.forward")


# ---------------------------------------------------------------------------------------------
# the network function
# ---------------------------------------------------------------------------------------------
class _ResCNNTrainFn(torch.autograd.Function):
    """Train-mode forward/backward of the whole embedding network as one autograd node."""

    @staticmethod
    def forward(ctx, x, model, *params):
        eng = get_engine()
        prec = model._train_arith(x)
        if prec == "f16":
            from .train_f16 import forward_train_group_f16
            pw = model._packed(with_dgrad=True, with_f16=True, f32_banks=False, with_f16_dgrad=True)
            (e,), saved = forward_train_group_f16(eng, [x], pw, model._bn_params(), save=True, reducer=model._reducer)
        else:
            pw = model._packed(with_dgrad=True, with_bf16=(prec == "bf16x3"), f32_banks=(prec != "bf16x3"))
            e, saved = eng.forward_train(x, pw, model._bn_params(), save=True, reducer=model._reducer, precision=prec)
        ctx.precision = prec
        model._bump_batches_tracked(1)        # nn.BatchNorm2d.train() bookkeeping, one launch
        model._stat_updates += 1
        ctx.saved_forward = saved
        ctx.model = model
        ctx.pw = pw
        ctx.param_names = model._param_names
        return e

    @staticmethod
    def backward(ctx, ge):
        from .backward import backward_train
        bn_w = {n: m.weight for n, m in zip(ctx.model._bn_names(), ctx.model._bn_modules())}
        if ctx.precision == "f16":
            from .train_f16 import backward_train_f16
            grads = backward_train_f16(get_engine(), bn_w, ctx.pw, ctx.saved_forward, ge.contiguous().float(),
                                       loss_scale=ctx.model.loss_scale, reducer=ctx.model._reducer,
                                       reduce_gradients=ctx.model._reducer is not None,
                                       overflow_flag=ctx.model.grad_overflow_flag(ge.device))
        else:
            grads = backward_train(get_engine(), bn_w, ctx.pw, ctx.saved_forward, ge.contiguous().float(),
                                   reducer=ctx.model._reducer, precision=ctx.precision,
                                   reduce_gradients=ctx.model._reducer is not None)
        ctx.saved_forward = None
        return (None, None) + tuple(grads.get(n) for n in ctx.param_names)


class _ResCNNTripletFn(torch.autograd.Function):
    """The three train-mode forwards of a triplet step (train_triplet.py:215) and their backward as ONE autograd node
    over the concatenated batch (Engine.forward_train_group)."""

    @staticmethod
    def forward(ctx, xa, xp, xn, model, *params):
        from .engine import trace_range
        eng = get_engine()
        prec = model._train_arith(xa, members=3)
        with trace_range(f"ds.train.forward_triplet[{prec}]"):
            if prec == "f16":
                from .train_f16 import forward_train_group_f16
                pw = model._packed(with_dgrad=True, with_f16=True, f32_banks=False, with_f16_dgrad=True)
                embs, saved = forward_train_group_f16(eng, [xa, xp, xn], pw, model._bn_params(), save=True,
                                                      reducer=model._reducer)
            else:
                pw = model._packed(with_dgrad=True, with_bf16=(prec == "bf16x3"), f32_banks=(prec != "bf16x3"))
                embs, saved = eng.forward_train_group([xa, xp, xn], pw, model._bn_params(), save=True, reducer=model._reducer,
                                                      precision=prec)
        model._bump_batches_tracked(3)        # nn.BatchNorm2d.train() bookkeeping, one launch
        model._stat_updates += 3
        ctx.precision, ctx.saved_forward, ctx.model, ctx.pw = prec, saved, model, pw
        ctx.param_names = model._param_names
        return tuple(e.clone() for e in embs)

    @staticmethod
    def backward(ctx, ga, gp, gn):
        from .backward import backward_train
        ref = next(g for g in (ga, gp, gn) if g is not None)
        ge = torch.cat([g if g is not None else torch.zeros_like(ref) for g in (ga, gp, gn)]).contiguous().float()
        bn_w = {n: m.weight for n, m in zip(ctx.model._bn_names(), ctx.model._bn_modules())}
        from .engine import trace_range
        with trace_range(f"ds.train.backward_triplet[{ctx.precision}]"):
            if ctx.precision == "f16":
                from .train_f16 import backward_train_f16
                grads = backward_train_f16(get_engine(), bn_w, ctx.pw, ctx.saved_forward, ge, loss_scale=ctx.model.loss_scale,
                                           reducer=ctx.model._reducer, reduce_gradients=ctx.model._reducer is not None,
                                           overflow_flag=ctx.model.grad_overflow_flag(ge.device))
            else:
                grads = backward_train(get_engine(), bn_w, ctx.pw, ctx.saved_forward, ge, reducer=ctx.model._reducer,
                                       precision=ctx.precision, reduce_gradients=ctx.model._reducer is not None)
        ctx.saved_forward = None
        return (None, None, None, None) + tuple(grads.get(n) for n in ctx.param_names)


class GraphedEmbedder:
    """The eval forward of one input shape captured once into a HIP graph and replayed per call.

    `emb = g(x)` copies x into the graph's static input, replays, and returns the graph's static output
    tensor (valid until the next call; clone it to keep it).  Weights are read at replay time through the
    packed copies made at capture: re-capture after changing parameters."""

    def __init__(self, model: "DeepSpeakerModel", example: torch.Tensor):
        _require_cuda(example, "GraphedEmbedder")
        if model.training:
            raise RuntimeError("GraphedEmbedder captures the eval forward: call model.eval() first")
        self.model = model
        self.static_x = example.detach().contiguous().float().clone()
        cur = torch.cuda.current_stream(example.device)
        side = torch.cuda.Stream(device=example.device)
        side.wait_stream(cur)
        with torch.no_grad(), torch.cuda.stream(side):     # warm-up off the capture: launch plans, LDS opt-ins
            for _ in range(3):
                model(self.static_x)
        cur.wait_stream(side)
        torch.cuda.synchronize(example.device)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.static_e = model(self.static_x)

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        if tuple(x.shape) != tuple(self.static_x.shape):
            raise ValueError(f"captured for inputs of shape {tuple(self.static_x.shape)}, got {tuple(x.shape)}")
        self.static_x.copy_(x)
        self.graph.replay()
        return self.static_e


class DeepSpeakerModel(nn.Module):
    """reference model.py:153-223.

    `DeepSpeakerModel(embedding_size, num_classes, feature_dim=64)`; `.forward(x[B,1,T,64]) -> [B,512]`
    L2-normalised x10 embedding (also cached on `.features`, model.py:210-213);
    `.forward_classifier(x)`; `.l2_norm(t)`.  `state_dict()` has the reference's 76 keys.
    """

    def __init__(self, embedding_size, num_classes, feature_dim=64, n_stages: int = 4, precision: str = "f32",
                 low_latency: bool = False, train_precision: Optional[str] = None, loss_scale: float = 1024.0,
                 f16_guard: Optional[float] = F16_GUARD_THRESHOLD):
        super().__init__()
        # precision "f16": the eval path measures its own distance to the f32-class path on sample rows and runs the
        # f32-class kernels instead while that estimate is above `f16_guard` (precision_guard.py; None: never)
        object.__setattr__(self, "f16_guard", F16Guard(f16_guard) if (f16_guard is not None and precision == "f16") else None)
        # arithmetic of TRAINING steps: None = the f32-class default ("bf16x3" when `precision` is a 16-bit one, else
        # "f32"); "f16" = the opt-in fp16 step (train_f16.py: fp16 activations and loss-scaled fp16 gradients in HBM, one
        # fp16 MFMA per product; embeddings / loss within 1e-3, gradients within 3e-3 of the masked oracle)
        if train_precision not in (None, "f32", "bf16x3", "f16"):
            raise ValueError(f"unknown train_precision {train_precision!r}; expected None, 'f32', 'bf16x3' or 'f16'")
        self.train_precision = train_precision
        self.loss_scale = float(loss_scale)
        # serving: eval forwards of a few utterances (fp16 path) split each layer's contraction over several
        # workgroups instead of letting a handful of workgroups walk it alone (results then depend on the batch
        # size in the last bits: the f32 summation order changes)
        self.low_latency = low_latency
        # arithmetic of the stage convolutions: "f32" (exact-f32 MFMA), "bf16x3" (split-operand bf16 MFMA,
        # f32-class accuracy; in training: forward, data and filter gradients of the 3x3 / 5x5 layers; the fc
        # layer, conv1's filter gradient and the BatchNorm / loss passes stay f32), "f16" (eval: fp16 operands
        # and activations, f32 accumulate -- the throughput path; training then runs in bf16x3) or "bf16"
        # (eval-only speed mode outside the 1e-3 contract; training then runs in f32)
        if precision not in PRECISIONS:
            raise ValueError(f"unknown precision {precision!r}; expected one of {PRECISIONS}")
        self.precision = precision
        if feature_dim != 64:
            # the reference's feature_dim == 40 branch is dead code that cannot run
            raise NotImplementedError("only feature_dim=64 is functional in the reference and implemented here")
        self.embedding_size = embedding_size
        self.n_stages = n_stages
        self.model = myResNet(BasicBlock, [1, 1, 1, 1], n_stages=n_stages)
        c_last = STAGE_CHANNELS[n_stages - 1]
        f_bins = 64 >> n_stages
        self.model.fc = nn.Linear(c_last * f_bins, self.embedding_size)       # 512*4 (model.py:164)
        self.model.classifier = nn.Linear(self.embedding_size, num_classes)   # model.py:167
        self._reducer = None          # set by enable_data_parallel()
        self._pack_cache = {}
        self._pack_key = None
        self._stat_updates = 0        # train-mode forwards so far (invalidates the folded BatchNorm cache)
        self._fold_cache = None
        self._fold_key = None
        self._param_names = [n for n, _ in self.named_parameters() if not n.startswith("model.classifier")]

    # ---- caches of derived tensors, invalidated by parameter version counters ----
    def _bn_modules(self):
        out = []
        for i in range(1, self.n_stages + 1):
            blk = getattr(self.model, f"layer{i}")[0]
            out += [getattr(self.model, f"bn{i}"), blk.bn1, blk.bn2]
        return out

    def _bump_batches_tracked(self, n: int):
        """nn.BatchNorm2d.train() bookkeeping for `n` forwards: every BatchNorm layer's batch counter goes up by n in
        ONE launch -- the twelve counters are kept as views of one int64 tensor (re-established whenever .to() /
        .cuda() has given the modules separate buffers again; load_state_dict copies in place and keeps the views)."""
        mods = self._bn_modules()
        flat = getattr(self, "_nbt_flat", None)
        first = mods[0].num_batches_tracked
        # every module's counter must still be ITS view (load_state_dict(assign=True), a manual buffer replacement or a
        # partial _apply can re-assign any one of them; host-side pointer compares, no device work)
        if (flat is None or flat.device != first.device
                or any(m.num_batches_tracked.data_ptr() != flat.data_ptr() + 8 * i for i, m in enumerate(mods))):
            flat = torch.stack([m.num_batches_tracked.reshape(()) for m in mods]).to(torch.int64)
            for i, m in enumerate(mods):
                m._buffers["num_batches_tracked"] = flat[i]
            object.__setattr__(self, "_nbt_flat", flat)
        flat.add_(n)

    def _bn_names(self):
        out = []
        for i in range(1, self.n_stages + 1):
            out += [f"model.bn{i}", f"model.layer{i}.0.bn1", f"model.layer{i}.0.bn2"]
        return out

    def _bn_params(self) -> Dict[str, BNParams]:
        return {n: BNParams(m.weight, m.bias, m.running_mean, m.running_var)
                for n, m in zip(self._bn_names(), self._bn_modules())}

    def _conv_fc_tensors(self):
        sd = {}
        for i in range(1, self.n_stages + 1):
            blk = getattr(self.model, f"layer{i}")[0]
            sd[f"model.conv{i}.weight"] = getattr(self.model, f"conv{i}").weight
            sd[f"model.layer{i}.0.conv1.weight"] = blk.conv1.weight
            sd[f"model.layer{i}.0.conv2.weight"] = blk.conv2.weight
        sd["model.fc.weight"] = self.model.fc.weight
        sd["model.fc.bias"] = self.model.fc.bias
        return sd

    # ---- the loss-scaled fp16 training step's overflow handling (train_precision="f16") ----
    def grad_overflow_flag(self, device=None) -> torch.Tensor:
        """int32 device tensor [1]: raised (set to 1, never cleared) by every backward pass of the fp16 training step whose
        scaled gradients left fp16's range (inf / NaN in a filter gradient).  Passes only OR into it -- the reference's
        canonical step is three `model(x)` calls, i.e. three backward passes per `loss.backward()`, and so is gradient
        accumulation -- and the fused optimizers of `optim.create_optimizer(model, ...)` CONSUME it: they read it on the
        device, skip the update while it is set (no host synchronisation), then latch it into `grad_overflow` and clear
        it (`_consume_overflow`).  `loss_scale` itself is static unless the training loop calls `update_loss_scale()`."""
        dev = torch.device(device if device is not None else next(self.parameters()).device)
        st = self.__dict__.get("_overflow_state")
        if st is None or st.device != dev:
            if dev.type == "cuda" and dev.index is None:
                dev = torch.device("cuda", torch.cuda.current_device())
            if st is None or st.device != dev:
                st = torch.zeros(2, dtype=torch.int32, device=dev)      # [pending, latched by the last optimizer step]
                object.__setattr__(self, "_overflow_state", st)
                object.__setattr__(self, "_overflow_flag", st[0:1])
        return self.__dict__["_overflow_flag"]

    def _consume_overflow(self):
        """optim._FusedBase.step, after its last launch: latch the pending flag for `grad_overflow` / `update_loss_scale`
        and clear it for the next step's passes (two 4-byte device operations on the caller's stream)."""
        st = self.__dict__.get("_overflow_state")
        if st is not None:
            st[1:2].copy_(st[0:1])
            st[0:1].zero_()

    @property
    def grad_overflow(self) -> bool:
        """Did the backward passes of the last optimizer step (or the passes since it) overflow?  (Reads the flags: a
        host synchronisation.)"""
        st = self.__dict__.get("_overflow_state")
        return bool(st.max().item()) if st is not None else False

    def update_loss_scale(self, backoff: float = 0.5, growth: float = 2.0, growth_interval: int = 2000,
                          max_scale: float = 65536.0) -> bool:
        """Dynamic loss scaling for loops that want it (torch.amp.GradScaler.update's rule): call after optimizer.step().
        Overflow in the last pass -> loss_scale *= backoff (that step was skipped on the device); `growth_interval` clean
        passes in a row -> loss_scale *= growth.  One host synchronisation per call; returns whether the step overflowed."""
        bad = self.grad_overflow
        if bad:
            self.loss_scale = max(1.0, self.loss_scale * backoff)
            self._clean_steps = 0
        else:
            self._clean_steps = self.__dict__.get("_clean_steps", 0) + 1
            if self._clean_steps >= growth_interval:
                self.loss_scale = min(max_scale, self.loss_scale * growth)
                self._clean_steps = 0
        return bad

    def _train_arith(self, x: Optional[torch.Tensor] = None, members: int = 1) -> str:
        """arithmetic of the next training step (see `train_precision`); `x`: one member's input batch"""
        tp = self.train_precision
        if tp is None:
            return "bf16x3" if self.precision in ("bf16x3", "f16") else "f32"
        if tp == "f16" and x is not None:
            # the fp16 BatchNorm backward addresses the parity-class layout of the stride-2 data gradient with 24-bit
            # pixel indices (csrc/train_f16.hip bwd_reduce_f16): all members' stage-1 pixels must stay below 2^24 (the
            # bench step has 1.97 M).  Larger steps run the f32-class arithmetic instead of failing inside the pass.
            pixels = members * x.shape[0] * ((x.shape[2] - 1) // 2 + 1) * 32
            if pixels >= (1 << 24):
                if not self.__dict__.get("_warned_f16_step_size"):
                    import warnings
                    warnings.warn(f"fp16 training step: {pixels} stage-1 pixels in one step exceed the 2^24 of its BatchNorm "
                                  "backward; this step (and every one this large) runs train_precision='bf16x3'")
                    self.__dict__["_warned_f16_step_size"] = True
                return "bf16x3"
        return tp

    def _packed(self, with_dgrad: bool = False, with_bf16: bool = False, with_f16: bool = False, f32_banks: bool = True,
                with_f16_dgrad: bool = False):
        """Kernel-layout copies of the filters for one set of consumers; one copy per variant is kept until a
        parameter changes (version counters), so alternating precisions do not re-pack."""
        sd = self._conv_fc_tensors()
        key = tuple((t.data_ptr(), t._version) for t in sd.values())
        if self._pack_key != key:
            self._pack_cache, self._pack_key = {}, key
        variant = (with_dgrad, with_bf16, with_f16, f32_banks, with_f16_dgrad)
        pw = self._pack_cache.get(variant)
        if pw is None:
            pw = self._pack_cache[variant] = get_engine().pack_weights(sd, self.n_stages, with_dgrad=with_dgrad,
                                                                       with_bf16=with_bf16, with_f16=with_f16,
                                                                       f32_banks=f32_banks, with_f16_dgrad=with_f16_dgrad)
            pw.owner = id(self)         # lets the engine's plan cache drop this model's older generations
        return pw

    def _folded(self):
        mods = self._bn_modules()
        # num_batches_tracked (a host-visible counter bumped by every train-mode forward) stands in for the
        # running statistics' version: the finalize kernels update those through raw pointers
        key = tuple((t.data_ptr(), t._version) for m in mods
                    for t in (m.weight, m.bias, m.running_mean, m.running_var)) + (self._stat_updates,)
        if self._fold_key != key:
            eng = get_engine()
            self._fold_cache = {n: eng.bn_fold(b) for n, b in self._bn_params().items()}
            self._fold_key = key
        return self._fold_cache

    # ---- data-parallel training (new capability; the reference is single-GPU) ----
    def enable_data_parallel(self, process_group=None, force: bool = False, grad_comm=None, grad_reduce=None,
                             grad_group=None):
        """One process per GPU (torch.distributed backend "nccl" = RCCL).  From now on train-mode forwards and
        their backward use global-batch BatchNorm statistics (all-reduced over the ranks -- one collective per
        BatchNorm layer when the step goes through `forward_triplet`), and the backward pass all-reduces the
        filter / fc gradients itself, one bucket per stage, launched as soon as the stage's gradient kernels are
        enqueued (overlapped with the rest of the pass).  Scale the local loss by 1/world_size so that the step
        equals the single-process step on the concatenated batch; the classifier head (not part of the embedding
        network's autograd node) is reduced by `allreduce_gradients()`."""
        from .distributed import Reducer
        # force: run the data-parallel launch sequence (collectives included) even in a group of one -- the way the
        # path is exercised and timed on a single GPU (bench.py --force-collectives)
        # grad_comm / grad_reduce / grad_group: how the gradient buckets travel (distributed.Reducer)
        self._reducer = Reducer(process_group, force=force, grad_comm=grad_comm, grad_reduce=grad_reduce,
                                grad_group=grad_group)
        return self._reducer

    def allreduce_gradients(self):
        """Sum over the ranks what the backward pass has not already reduced: the classifier head's gradients
        (the embedding network's filter / fc gradients are all-reduced inside its backward; BatchNorm affine
        gradients come out of global sums)."""
        from .distributed import allreduce_gradients
        if self._reducer is None:
            raise RuntimeError("call enable_data_parallel() first")
        grads = {n: p.grad for n, p in self.named_parameters()
                 if p.grad is not None and n.startswith("model.classifier")}
        if grads:
            allreduce_gradients(grads, self._reducer, n_buckets=1)

    # ---- reference surface ----
    def l2_norm(self, input):
        """reference model.py:172-183: x / sqrt(sum x^2 + 1e-10) (no alpha)."""
        _require_cuda(input, "l2_norm")
        eng = get_engine()
        x = input.contiguous().view(input.size(0), -1)
        out = torch.empty_like(x)
        eng.lib.call("ds_l2norm_scale_f32", eng._p(x), eng._p(out), x.shape[0], x.shape[1], 1.0, L2_EPS,
                     eng._stream(x))
        return out.view(input.size())

    def forward(self, x):
        _require_cuda(x, "DeepSpeakerModel.forward")
        if x.dim() != 4 or x.size(1) != 1 or x.size(3) != 64:
            raise ValueError(f"expected input [B,1,T,64] (reference model.py:185; SURVEY F1), got {tuple(x.shape)}")
        x = x.contiguous().float()
        if self.training:
            params = [p for n, p in self.named_parameters() if not n.startswith("model.classifier")]
            if torch.is_grad_enabled() and any(p.requires_grad for p in params):
                self.features = _ResCNNTrainFn.apply(x, self, *params)
            else:
                e, _ = get_engine().forward_train(x, self._packed(), self._bn_params(), save=False,
                                                  reducer=self._reducer)
                self._bump_batches_tracked(1)        # nn.BatchNorm2d.train() bookkeeping, one launch
                self._stat_updates += 1
                self.features = e
        else:
            prec = self.eval_precision(x)
            pw = self._packed(with_bf16=prec in ("bf16x3", "bf16"), with_f16=prec == "f16")
            self.features = get_engine().forward_eval_planned(x, pw, self._folded(), precision=prec,
                                                              low_latency=self.low_latency)
        return self.features

    def _apply(self, fn, *args, **kwargs):
        """`.cuda()` / `.to(device)`: the library owns no device memory, so the zeroed buffer the persistent fp16 kernels draw
        their tile counters from is allocated here, through torch's allocator, for every device the model now lives on --
        eagerly, so that a stream capture whose first forward was never warmed up still finds its slots."""
        result = super()._apply(fn, *args, **kwargs)
        devices = {p.device for p in self.parameters() if p.is_cuda}
        if devices and not torch.cuda.is_current_stream_capturing():
            lib = get_engine().lib
            for dev in devices:
                lib.ensure_sched_workspace(dev)
        return result

    # new weights / a model coming back from training re-arm the fp16 path's precision check (precision_guard.py)
    def load_state_dict(self, *args, **kwargs):
        result = super().load_state_dict(*args, **kwargs)
        guard = self.__dict__.get("f16_guard")
        if guard is not None:
            guard.invalidate()
        return result

    def train(self, mode: bool = True):
        was_training = self.training
        super().train(mode)
        guard = self.__dict__.get("f16_guard")
        if guard is not None and was_training and not mode:
            guard.invalidate()
        return self

    def eval_precision(self, x: torch.Tensor, lengths=None) -> str:
        """The arithmetic the next eval forward of `x` runs in: `self.precision`, except that a model of precision "f16"
        whose guard has measured an fp16 error estimate above its threshold runs "bf16x3" (precision_guard.F16Guard;
        the first call on new weights measures -- one host synchronisation)."""
        if self.f16_guard is None or self.precision != "f16":
            return self.precision
        return self.f16_guard.precision_for(self, x, lengths)

    def calibrate_precision(self, x: torch.Tensor) -> Optional[dict]:
        """Run the fp16 guard's check on the first rows of `x` now (what the first eval forward on new weights does by
        itself; needed explicitly only before capturing the forward into a graph).  Returns the guard's report."""
        _require_cuda(x, "DeepSpeakerModel.calibrate_precision")
        if self.f16_guard is None:
            return None
        self.f16_guard.calibrate(self, x.contiguous().float())
        return self.f16_guard.report()

    def forward_triplet(self, data_a, data_p, data_n):
        """`out_a, out_p, out_n = model(data_a), model(data_p), model(data_n)` (train_triplet.py:215) as one call.
        Train mode: the three forwards run in lock-step over one concatenated batch with one BatchNorm statistic set
        per member (same values as three calls, three running-statistics updates in the same order), the backward
        pass is one pass over that batch, and data-parallel training exchanges one all-reduce per BatchNorm layer.
        Eval mode: one forward of the concatenated batch."""
        for t in (data_a, data_p, data_n):
            _require_cuda(t, "DeepSpeakerModel.forward_triplet")
            if t.dim() != 4 or t.size(1) != 1 or t.size(3) != 64 or t.shape != data_a.shape:
                raise ValueError(f"expected three equally shaped [B,1,T,64] batches, got {tuple(t.shape)}")
        xs = [t.contiguous().float() for t in (data_a, data_p, data_n)]
        params = [p for n, p in self.named_parameters() if not n.startswith("model.classifier")]
        if not self.training:
            e = self.forward(torch.cat(xs))
            return tuple(e.split(xs[0].shape[0]))
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            outs = _ResCNNTripletFn.apply(xs[0], xs[1], xs[2], self, *params)
        else:
            embs, _ = get_engine().forward_train_group(xs, self._packed(), self._bn_params(), save=False,
                                                       reducer=self._reducer)
            self._bump_batches_tracked(3)        # nn.BatchNorm2d.train() bookkeeping, one launch
            self._stat_updates += 3
            outs = tuple(embs)
        self.features = outs[2]
        return outs

    def embed_variable_length(self, utterances, max_batch: int = 2048, pad_to: int = 16, max_frames: int = 262144,
                              batch_step: int = 32, in_flight: int = 1):
        """Eval-mode embeddings of utterances of DIFFERENT lengths (BASELINE configs[4]: 100-800 frames; the
        temporal mean pool of model.py:207 accepts any T).  `utterances`: a sequence of [T_i, 64]
        (or [1, T_i, 64]) float tensors on the device, or a `data.FeatureStore` (the resident corpus: batches are
        then one gather kernel each).  They are sorted by length and packed into zero-padded batches whose length is
        the longest member's rounded up to `pad_to` frames (so launch plans are re-used): as many utterances as fit
        `max_frames` padded frames (about two 768 x 160-frame forwards: short utterances travel in larger batches, the
        GPU sees the same amount of work per launch) and at most `max_batch`.  They run through the masked forward:
        each embedding is bit-identical to the utterance's own forward -- padding never leaks
        (Engine.forward_eval_planned(lengths=...)).  `in_flight` > 1: consecutive batches alternate over that many HIP
        streams (one batch's assembly, HBM-bound first layer and small tail launches beside the other's matrix kernels:
        see pipeline.BatchesInFlight; +2 % at 2); the caller's stream is ordered after all of them before the result is
        returned.  Default 1: extra streams shift which hardware queue every stream created later lands on.
        Returns [N, embedding_size] in the order given."""
        if self.training:
            raise RuntimeError("embed_variable_length is an inference path: call model.eval() first")
        n = len(utterances)
        if n == 0:
            raise ValueError("no utterances")
        store = utterances if hasattr(utterances, "crops") else None      # a data.FeatureStore: one gather per batch
        feats = []
        if store is None:
            for u in utterances:
                _require_cuda(u, "DeepSpeakerModel.embed_variable_length")
                u = u.reshape(-1, u.shape[-1])
                if u.shape[1] != 64 or u.shape[0] < 1:
                    raise ValueError(f"expected [T, 64] utterances, got {tuple(u.shape)}")
                feats.append(u)
            lens = torch.tensor([u.shape[0] for u in feats], dtype=torch.int64)
            dev = feats[0].device
        else:
            lens = torch.tensor([store.length(i) for i in range(n)], dtype=torch.int64)
            dev = store.features.device
        order = torch.argsort(lens, stable=True)
        out = torch.empty((n, self.embedding_size), dtype=torch.float32, device=dev)
        eng = get_engine()
        sorted_lens = lens[order].tolist()
        # the arithmetic (precision "f16": the guard's verdict; on new weights it measures first, on a zero-padded batch
        # of up to 32 utterances around the median length) and every derived tensor -- packed filters, folded BatchNorm --
        # are settled HERE, on the caller's stream, before the lanes fork: built inside one lane, the other lanes' first
        # batches would read them with no ordering after the kernels that write them
        prec = self.precision
        if self.f16_guard is not None and prec == "f16":
            mid = order[max(0, n // 2 - 16):n // 2 + 16]
            ln_s = lens[mid]
            t_s = int(-(-int(ln_s.max()) // pad_to) * pad_to)
            if store is not None:
                x_s = store.crops(mid.numpy(), [0] * len(mid), t_s)
            else:
                x_s = torch.zeros((len(mid), 1, t_s, 64), dtype=torch.float32, device=dev)
                for r, j in enumerate(mid.tolist()):
                    x_s[r, 0, :feats[j].shape[0]].copy_(feats[j])
            prec = self.f16_guard.precision_for(self, x_s, ln_s)
        pw = self._packed(with_bf16=prec in ("bf16x3", "bf16"), with_f16=prec == "f16")
        folded = self._folded()
        order_dev = order.to(dev)                # once: a per-batch pageable copy would stall the host on the stream
        on_gpu = dev.type == "cuda"              # (the host emulator runs CPU tensors in program order)
        if not on_gpu:
            in_flight = 1
        caller = torch.cuda.current_stream(dev) if on_gpu else None
        lanes = [caller]
        if in_flight > 1:
            lanes = self.__dict__.setdefault("_varlen_lanes", {}).get(dev)
            if lanes is None or len(lanes) != in_flight:
                lanes = self.__dict__["_varlen_lanes"][dev] = [torch.cuda.Stream(device=dev) for _ in range(in_flight)]
            for lane in lanes:
                lane.wait_stream(caller)        # the utterances (and `out`) were produced on the caller's stream
        i = k = 0
        while i < n:
            # the largest count whose padded frames fit the budget (lengths ascend: the last member is the longest)
            cnt = 1
            while (cnt < max_batch and i + cnt < n
                   and (cnt + 1) * (-(-sorted_lens[i + cnt] // pad_to) * pad_to) <= max_frames):
                cnt += 1
            if cnt >= 2 * batch_step and i + cnt < n:
                cnt -= cnt % batch_step         # batch sizes in steps: (B, t_pad) launch plans repeat instead of being
            idx = order[i:i + cnt]              # built (and their buffers pinned) once per batch
            i += cnt
            ln = lens[idx]
            t_pad = int(-(-int(ln.max()) // pad_to) * pad_to)
            with torch.cuda.stream(lanes[k % len(lanes)]):      # (a no-op context for None)
                if store is not None:   # whole utterances from the resident corpus, zero-padded by the gather kernel
                    x = store.crops(idx.numpy(), [0] * len(idx), t_pad)
                else:
                    x = torch.zeros((len(idx), 1, t_pad, 64), dtype=torch.float32, device=dev)
                    for r, j in enumerate(idx.tolist()):
                        x[r, 0, :feats[j].shape[0]].copy_(feats[j])
                e = eng.forward_eval_planned(x, pw, folded, precision=prec, lengths=ln)
                out.index_copy_(0, order_dev[i - cnt:i], e)
            k += 1
        if in_flight > 1:
            for lane in lanes:
                caller.wait_stream(lane)
            out.record_stream(lanes[0])
        return out

    def embed_reference(self, x: torch.Tensor) -> torch.Tensor:
        """Eval-mode embeddings at f32-class precision (split-operand bf16 path) whatever `self.precision` is:
        what the near-tie refinement of `mining.select_triplets` re-embeds with."""
        _require_cuda(x, "DeepSpeakerModel.embed_reference")
        prec = "f32" if self.precision == "f32" else "bf16x3"
        pw = self._packed(with_bf16=prec == "bf16x3")
        return get_engine().forward_eval_planned(x.contiguous().float(), pw, self._folded(), precision=prec)

    def graphed(self, example: torch.Tensor) -> "GraphedEmbedder":
        """HIP-graph replay of the eval forward for inputs of `example`'s shape.  New capability; the reference has no
        counterpart.  (Measured on MI355X / ROCm 7: the eager launch plan already keeps the GPU's queue full -- host
        enqueue is ~0.3 ms for a forward whose kernels take longer even at B = 1 -- so replay is no faster, 249 vs
        238 us; it removes the host work, which matters when the host is busy with something else.)"""
        return GraphedEmbedder(self, example)

    def _head(self) -> HeadPack:
        w, b = self.model.classifier.weight, self.model.classifier.bias
        key = (w.data_ptr(), w._version, b.data_ptr(), b._version)
        if getattr(self, "_head_key", None) != key:
            self._head_cache, self._head_key = HeadPack(get_engine(), w, b), key
        return self._head_cache

    def classify(self, embeddings):
        """logits = classifier(embeddings) for embeddings that already exist: the softmax pre-training regime
        (train_triplet.py:277-279) re-runs the whole network three times through `forward_classifier` to classify
        the very rows `model(data_a)` etc. produced a moment earlier; `classify(out_a[idx])` re-uses them."""
        _require_cuda(embeddings, "DeepSpeakerModel.classify")
        return _LinearHeadFn.apply(embeddings, self.model.classifier.weight, self.model.classifier.bias, get_engine(),
                                   self._head())

    def classifier_loss(self, embeddings, labels):
        """CrossEntropyLoss(classifier(embeddings), labels) (train_triplet.py:281-285, mean reduction) as one fused
        GEMM + log-softmax + NLL; gradients flow to the classifier and into `embeddings`."""
        _require_cuda(embeddings, "DeepSpeakerModel.classifier_loss")
        return _HeadCrossEntropyFn.apply(embeddings, self.model.classifier.weight, self.model.classifier.bias, labels,
                                         get_engine(), self._head())

    def forward_classifier(self, x):
        """reference model.py:220-223: logits = classifier(embedding x10), on the f32 matrix cores."""
        return self.classify(self.forward(x))
