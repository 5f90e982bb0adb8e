"""The OPT-IN fp16 training step: `DeepSpeakerModel(..., train_precision="f16")`.

Same function as the f32-class step (`Engine.forward_train_group` + `backward.backward_train`, i.e. the reference's
`out_a, out_p, out_n = model(data_a), model(data_p), model(data_n)`; `loss.backward()` -- train_triplet.py:215-223), other
arithmetic and other bytes:

* every activation, pre-activation and gradient tensor is fp16 in HBM (the network input, the last stage's output and
  everything behind it -- pooling, projection, l2-norm, loss -- stay f32);
* forward convolutions AND data gradients run on the eval path's fp16 matrix-core kernels (`ds_conv_fwd_f16`: persistent
  workgroups, one MFMA per product, f32 accumulate) -- a 3x3 data gradient is that kernel over dL/dz with the flipped
  bank, a 5x5 stride-2 data gradient is ONE 3x3 convolution whose 4 Cin output channels are the four parity classes of
  dX (read in place by the BatchNorm backward of the layer below);
* BatchNorm statistics come from a pass of their own over the fp16 pre-activation, BatchNorm / clip passes move half
  the bytes of the f32 step's; filter gradients contract fp16 operands (`ds_conv_wgrad_f16`), f32 accumulate;
* gradient tensors hold `loss_scale` * g (a constant power of two; the small gradients of the early layers then sit in
  fp16's normal range); parameter gradients leave in f32, un-scaled.

Stated tolerance (tests/test_gpu_train_f16.py; measured at the 768-row bench batch in brackets): loss within 1e-3 (3.5e-4)
and train-mode embeddings within 2e-3 (1.15e-3) of the reference's recorded step, every parameter gradient within 8e-3
rel-L2 (4.0e-3 worst, 1.4e-3 median) of the oracle evaluated with this forward's clip masks and dL/de -- the order of the
4e-3 the reference's own float32 run is from its float64 run.  The f32-class step (bf16x3, 1e-4) stays the default.

Data parallelism (`enable_data_parallel`): the same exchange pattern as the f32-class step -- one float64 all-reduce per
BatchNorm layer and direction carrying all members' sums, f32 gradient buckets per stage (tests/test_distributed_gloo.py).
"""
from __future__ import annotations

import ctypes
from typing import Dict, List, Optional

import torch

from ._native import ConvShape, DS_EPI_CLIP, DS_EPI_OUT_F16, DS_EPI_OUT_F32, DS_EPI_RESIDUAL
from .engine import ALPHA, BN_EPS, BN_MOMENTUM, L2_EPS, STAGE_CHANNELS, BNParams, Engine, PackedWeights, SavedForward, conv_out

DEFAULT_LOSS_SCALE = 1024.0


def _bn_stats(eng: Engine, z16: torch.Tensor, bn: BNParams, G: int, update_running: bool = True, reducer=None, tables=None):
    """[G][C] tables (mean, invstd, scale, shift) of a train-mode BatchNorm over the G members of z16 [B,h,w,C] fp16.
    With an active `reducer` (data parallelism) the members' float64 sums travel in ONE all-reduce, so every rank
    normalises with the statistics of the global batch."""
    c = z16.shape[-1]
    n_pix = (z16.numel() // c) // G
    rows = eng.lib.raw("ds_bn_f16_partial_rows")(n_pix, c)
    partial = torch.empty((G, rows, c, 2), dtype=torch.float32, device=z16.device)
    if tables is None:
        tables = torch.empty((4, G, c), dtype=torch.float32, device=z16.device)
    st = eng._stream(z16)
    if reducer is not None and reducer.active:
        sums = torch.empty((G, 2 * c + 1), dtype=torch.float64, device=z16.device)
        eng.lib.call("ds_bn_stats_partial_f16", eng._p(z16), eng._p(partial), n_pix, c, G, st)
        eng.lib.call("ds_partial_sum_f64_group", eng._p(partial), rows, eng._p(sums), n_pix, c, G, st)
        reducer.all_reduce_sum_(sums)
        eng.lib.call("ds_bn_stats_from_sums_group_f32", eng._p(sums), eng._p(bn.weight.detach()), eng._p(bn.bias.detach()),
                     BN_EPS, BN_MOMENTUM, eng._p(bn.running_mean) if update_running else None,
                     eng._p(bn.running_var) if update_running else None, eng._p(tables[0]), eng._p(tables[1]),
                     eng._p(tables[2]), eng._p(tables[3]), c, G, st)      # members in call order inside the kernel
        return tables
    eng.lib.call("ds_bn_stats_group_f16", eng._p(z16), eng._p(partial), n_pix, eng._p(bn.weight.detach()),
                 eng._p(bn.bias.detach()), BN_EPS, BN_MOMENTUM, eng._p(bn.running_mean) if update_running else None,
                 eng._p(bn.running_var) if update_running else None, eng._p(tables[0]), eng._p(tables[1]),
                 eng._p(tables[2]), eng._p(tables[3]), c, G, st)
    return tables


def _bn_apply(eng: Engine, z16, tables, residual16, G: int, flags: int, out=None):
    c = z16.shape[-1]
    n_pix = (z16.numel() // c) // G
    y = out if out is not None else torch.empty(z16.shape, dtype=torch.float32 if flags & DS_EPI_OUT_F32 else torch.float16,
                                                device=z16.device)
    eng.lib.call("ds_bn_apply_group_f16", eng._p(z16), eng._p(tables[2]), eng._p(tables[3]), eng._p(residual16), eng._p(y),
                 n_pix, c, G, flags, eng._stream(z16))
    return y


def forward_train_group_f16(eng: Engine, xs: List[torch.Tensor], pw: PackedWeights, bns: Dict[str, BNParams],
                            save: bool = True, reducer=None):
    """The train-mode forwards of the G = len(xs) members in lock-step over one concatenated batch, fp16 tensors between
    the layers.  One convolution launch per layer over ALL members (the statistics are a separate pass, so tiles may
    straddle members), one statistics launch pair, one normalise + clip launch.  (Measured and dropped: one HIP stream per
    member, as the f32-class forward runs -- bitwise the same results, 8.89 against 8.86 ms per step and 5.4 instead of 3.8
    ms of host enqueue: the fp16 passes are short enough that three times the launches cost what the overlap gains; the
    same for the backward chain, 9.9 ms.)  `reducer` (data parallelism): one
    float64 all-reduce per BatchNorm layer carrying all members' sums.  Returns ([embeddings per member],
    SavedForward with fp16 `raws` / `acts` -- the last stage's output is f32 -- and `stats[name]` = the [4][G][C] tables)."""
    G = len(xs)
    for x in xs:
        eng._check(x, "input")
        if x.shape != xs[0].shape or x.dim() != 4 or x.shape[1] != 1:
            raise ValueError("members must be equally shaped [B,1,T,F] batches")
    if pw.stages[0].l_conv1_f16 is None:
        raise ValueError("pack_weights(..., with_f16=True) is required for the fp16 training step")
    Bm, _, T, F = xs[0].shape
    B = G * Bm
    nbytes = xs[0].numel() * xs[0].element_size()
    if G == 1:
        x = xs[0]
    elif all(t.is_contiguous() and t.data_ptr() == xs[0].data_ptr() + g * nbytes
             and t.untyped_storage().data_ptr() == xs[0].untyped_storage().data_ptr() for g, t in enumerate(xs)):
        x = torch.as_strided(xs[0], (B, 1, T, F), xs[0].stride())
    else:
        x = torch.cat(xs)
    saved = SavedForward(x=x) if save else None
    n_stages = len(pw.stages)
    h, w, cin = T, F, 1
    a = x
    for s, sw in enumerate(pw.stages):
        i, c = s + 1, STAGE_CHANNELS[s]
        last = s == n_stages - 1
        if i == 1:      # Cin = 1: the split-operand kernel of the eval path, raw fp16 output
            z, _ = eng.conv1(a, sw.conv, B, h, w, flags=DS_EPI_OUT_F16, lowp=True)
        else:
            z = eng.conv_f16(a, sw.conv_f16, B, h, w, cin, c, 5, 2)
        h, w, cin = conv_out(h, 5, 2), conv_out(w, 5, 2), c
        name = f"model.bn{i}"
        tb = _bn_stats(eng, z, bns[name], G, reducer=reducer)
        a = _bn_apply(eng, z, tb, None, G, DS_EPI_CLIP)
        if save:
            saved.raws[name], saved.stats[name], saved.acts[f"stage{i}.a"] = z, tb, a
        name = f"model.layer{i}.0.bn1"
        z = eng.conv_f16(a, sw.l_conv1_f16, B, h, w, c, c, 3, 1)
        tb = _bn_stats(eng, z, bns[name], G, reducer=reducer)
        y = _bn_apply(eng, z, tb, None, G, DS_EPI_CLIP)
        if save:
            saved.raws[name], saved.stats[name], saved.acts[f"stage{i}.b"] = z, tb, y
        name = f"model.layer{i}.0.bn2"
        z = eng.conv_f16(y, sw.l_conv2_f16, B, h, w, c, c, 3, 1)
        tb = _bn_stats(eng, z, bns[name], G, reducer=reducer)
        a = _bn_apply(eng, z, tb, a, G, DS_EPI_CLIP | DS_EPI_RESIDUAL | (DS_EPI_OUT_F32 if last else 0))
        if save:
            saved.raws[name], saved.stats[name], saved.acts[f"stage{i}.c"] = z, tb, a
            saved.dims.append((h, w))
    e = eng.tail(a, pw, saved)
    return [e[g * Bm:(g + 1) * Bm] for g in range(G)], saved


def _bn_bwd(eng: Engine, g1, g1_parity, g2, act, z16, tables, gamma, G: int, hw, inv_scale: float, mask_from_z: bool = False,
            want_gy: bool = True, reducer=None):
    """(gy16 or None, gz16, dgamma, dbeta) of one BatchNorm + clip layer; see ds_bn_bwd_group_f16.  mask_from_z: the clip
    mask is re-derived from the pre-activation z16 and the forward's scale / shift tables instead of being read from a
    stored activation; want_gy=False (then also: no g2, no parity layout): the masked gradient is not stored."""
    c = z16.shape[-1]
    n_pix = (z16.numel() // c) // G
    dev = z16.device
    rows = eng.lib.raw("ds_bn_f16_partial_rows")(n_pix, c)
    partial = torch.empty((G, rows, c, 2), dtype=torch.float32, device=dev)
    coef = torch.empty((G, 3 * c), dtype=torch.float32, device=dev)
    gz = torch.empty_like(z16)
    gy = torch.empty_like(z16) if want_gy else None
    gg, gb = torch.empty(c, dtype=torch.float32, device=dev), torch.empty(c, dtype=torch.float32, device=dev)
    act_p = None if mask_from_z else eng._p(act)
    act32 = int(act is not None and act.dtype == torch.float32)
    msc, msh = (eng._p(tables[2]), eng._p(tables[3])) if mask_from_z else (None, None)
    st = eng._stream(z16)
    if reducer is not None and reducer.active:
        # the grouped launches split where the sums of ALL members travel in one all-reduce
        eng.lib.call("ds_bn_bwd_group_reduce_f16", eng._p(g1), int(g1_parity), eng._p(g2), act_p, act32, msc, msh, eng._p(z16),
                     eng._p(tables[0]), eng._p(tables[1]), eng._p(gy), eng._p(partial), n_pix, hw[0], hw[1], c, G, st)
        sums = torch.empty((G, 2 * c + 1), dtype=torch.float64, device=dev)
        eng.lib.call("ds_partial_sum_f64_group", eng._p(partial), rows, eng._p(sums), n_pix, c, G, st)
        reducer.all_reduce_sum_(sums)
        eng.lib.call("ds_bn_bwd_group_apply_f16", eng._p(sums), eng._p(gy if gy is not None else g1), int(gy is None), msc, msh,
                     eng._p(z16), eng._p(tables[0]), eng._p(tables[1]), eng._p(gamma.detach()), eng._p(coef), eng._p(gg),
                     eng._p(gb), eng._p(gz), n_pix, c, G, float(inv_scale), st)
        return gy, gz, gg, gb
    eng.lib.call("ds_bn_bwd_group_f16", eng._p(g1), int(g1_parity), eng._p(g2), act_p, act32, msc, msh, eng._p(z16),
                 eng._p(tables[0]), eng._p(tables[1]), eng._p(gamma.detach()), eng._p(gy), eng._p(partial), eng._p(coef),
                 eng._p(gg), eng._p(gb), eng._p(gz), n_pix, hw[0], hw[1], c, G, float(inv_scale), st)
    return gy, gz, gg, gb


def _wgrad(eng: Engine, shp: ConvShape, x16, gz16, out, inv_scale: float):
    n_ws = eng.lib.raw("ds_conv_wgrad_f16_workspace_floats")(ctypes.byref(shp))
    if n_ws <= 0:
        raise RuntimeError(f"ds_conv_wgrad_f16_workspace_floats failed: {n_ws}")
    ws = torch.empty(n_ws, dtype=torch.float32, device=x16.device)
    eng.lib.call("ds_conv_wgrad_f16", ctypes.byref(shp), eng._p(x16), eng._p(gz16), eng._p(ws), eng._p(out),
                 float(inv_scale), eng._stream(x16))
    return out


def _wgrad_c1(eng: Engine, shp: ConvShape, x32, gz16, out, inv_scale: float):
    n_ws = eng.lib.raw("ds_conv_wgrad_workspace_floats")(ctypes.byref(shp))
    ws = torch.empty(n_ws, dtype=torch.float32, device=x32.device)
    eng.lib.call("ds_conv_wgrad_c1_f16", ctypes.byref(shp), eng._p(x32), eng._p(gz16), eng._p(ws), eng._p(out),
                 float(inv_scale), eng._stream(x32))
    return out


def backward_train_f16(eng: Engine, bn_weights: Dict[str, torch.Tensor], pw: PackedWeights, saved: SavedForward,
                       ge: torch.Tensor, loss_scale: float = DEFAULT_LOSS_SCALE,
                       overlap_filter_gradients=None, reducer=None, reduce_gradients: bool = False,
                       overflow_flag: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
    """Parameter gradients (reference key names and shapes, f32, un-scaled) given dL/d(embedding) `ge` [B,512] f32, from
    the fp16 tensors `forward_train_group_f16` saved.  Filter gradients run on the second stream like the f32-class
    pass's (backward._FilterGradLane); the main chain stays on ONE stream over the whole batch (measured: every member's
    BatchNorm-backward / data-gradient chain on a stream of its own, as the forward does, is bitwise the same and SLOWER --
    9.9 against 8.9 ms per step: three times the launches, and 256-utterance convolutions fill the persistent grids
    worse than one 768-utterance launch).  `reducer` / `reduce_gradients` (data parallelism): global BatchNorm sums (one
    all-reduce per layer) and the per-stage f32 gradient buckets summed over the ranks, as backward.backward_train does.
    `overflow_flag` (int32 device tensor [1]): set to 1 at the end of the pass if any filter / fc gradient is inf or NaN
    (the static loss scale was too large for this step); never cleared and never read by the host here -- a step may be
    several passes (three `model(x)` calls, gradient accumulation), so the flag is cleared where it is CONSUMED: by the
    fused optimizer after its update (optim._FusedBase._consume_skip)."""
    from .backward import OVERLAP_FILTER_GRADIENTS, _FilterGradLane, _GradBuckets, _wgrad as _wgrad_f32
    lib = eng.lib
    lane = _FilterGradLane(ge.device, OVERLAP_FILTER_GRADIENTS if overlap_filter_gradients is None else overlap_filter_gradients)
    inv = 1.0 / float(loss_scale)
    grads: Dict[str, torch.Tensor] = {}
    n_stages = len(pw.stages)
    G = saved.stats["model.bn1"].shape[1]
    shapes = {n_stages: {"model.fc.weight": tuple(saved.fc_out.shape[1:]) + (saved.pooled.shape[1],),
                         "model.fc.bias": (saved.fc_out.shape[1],)}}
    for s_ in range(n_stages):
        i_, c_ = s_ + 1, STAGE_CHANNELS[s_]
        cin_ = 1 if s_ == 0 else STAGE_CHANNELS[s_ - 1]
        shapes[s_] = {f"model.layer{i_}.0.conv2.weight": (c_, c_, 3, 3), f"model.layer{i_}.0.conv1.weight": (c_, c_, 3, 3),
                      f"model.conv{i_}.weight": (c_, cin_, 5, 5)}
    buckets = _GradBuckets(shapes, ge.device, reducer if reduce_gradients else None)
    f = saved.fc_out
    B, n_out = f.shape
    st = eng._stream(f)
    # ---- f32 tail: l2-norm x alpha, fc, temporal mean + the last clip (model.py:205-213), as in backward_train ----
    gf = torch.empty_like(f)
    lib.call("ds_l2norm_scale_bwd_f32", eng._p(f), eng._p(ge), eng._p(gf), B, n_out, ALPHA, L2_EPS, st)
    pooled = saved.pooled
    k = pooled.shape[1]
    gb = buckets.views["model.fc.bias"]
    lib.call("ds_colsum_f32", eng._p(gf), eng._p(gb), B, n_out, st)
    grads["model.fc.bias"] = gb
    c_last = STAGE_CHANNELS[n_stages - 1]
    grads["model.fc.weight"] = _wgrad_f32(eng, ConvShape(1, B, 1, k, n_out, 1, 1), pooled, gf, (n_out, k), k // c_last,
                                          out=buckets.views["model.fc.weight"])
    buckets.done(n_stages)
    ws = torch.empty(lib.raw("ds_fc_workspace_floats")(B, n_out, k), dtype=torch.float32, device=f.device)
    gpooled = torch.empty((B, k), dtype=torch.float32, device=f.device)
    lib.call("ds_fc_l2norm_fwd_f32", eng._p(gf), eng._p(pw.fc_dgrad), None, eng._p(ws), eng._p(gpooled), None, B,
             n_out, k, 1.0, 0.0, st)
    out = saved.acts[f"stage{n_stages}.c"]                  # f32
    _, hr, wc, c = out.shape
    g32 = torch.empty_like(out)
    lib.call("ds_avgpool_time_bwd_f32", eng._p(gpooled), eng._p(out), eng._p(g32), B, hr, wc, c, st)
    g = torch.empty(out.shape, dtype=torch.float16, device=out.device)
    lib.call("ds_scale_cast_f32_to_f16", eng._p(g32), eng._p(g), g32.numel(), float(loss_scale), st)
    g_parity, g_masked = False, True
    for s in reversed(range(n_stages)):
        i, c = s + 1, STAGE_CHANNELS[s]
        h, w = saved.dims[s]
        cin = 1 if s == 0 else STAGE_CHANNELS[s - 1]
        a_act, b_act, c_act = (saved.acts[f"stage{i}.{t}"] for t in "abc")
        sw = pw.stages[s]
        shp3 = ConvShape(B, h, w, c, c, 3, 1)
        # out = clip(bn2(conv2(y)) + r)            (model.py:73-80)
        name = f"model.layer{i}.0.bn2"
        g_out, gz, gg, gbeta = _bn_bwd(eng, g, g_parity, None, None if g_masked else c_act, saved.raws[name],
                                       saved.stats[name], bn_weights[name], G, (h, w), inv, reducer=reducer)
        grads[name + ".weight"], grads[name + ".bias"] = gg, gbeta
        grads[f"model.layer{i}.0.conv2.weight"] = lane.run(
            lambda gz=gz: _wgrad(eng, shp3, b_act, gz, buckets.views[f"model.layer{i}.0.conv2.weight"], inv), gz)
        # y = clip(bn1(conv1(r)))                  (model.py:69-71)
        name = f"model.layer{i}.0.bn1"
        g_y = eng.conv_f16(gz, sw.l_conv2_dgrad_f16, B, h, w, c, c, 3, 1)
        # (no residual was added before this clip: its mask is re-derived from z; nobody else needs the masked gradient)
        _, gz, gg, gbeta = _bn_bwd(eng, g_y, False, None, None, saved.raws[name], saved.stats[name], bn_weights[name], G,
                                   (h, w), inv, mask_from_z=True, want_gy=False, reducer=reducer)
        grads[name + ".weight"], grads[name + ".bias"] = gg, gbeta
        grads[f"model.layer{i}.0.conv1.weight"] = lane.run(
            lambda gz=gz: _wgrad(eng, shp3, a_act, gz, buckets.views[f"model.layer{i}.0.conv1.weight"], inv), gz)
        # r = clip(bn_i(conv_i(x)));  dL/dr = conv path + residual path   (model.py:187-189, 67, 79)
        name = f"model.bn{i}"
        g_r = eng.conv_f16(gz, sw.l_conv1_dgrad_f16, B, h, w, c, c, 3, 1)
        _, gz, gg, gbeta = _bn_bwd(eng, g_r, False, g_out, None, saved.raws[name], saved.stats[name], bn_weights[name], G,
                                   (h, w), inv, mask_from_z=True, reducer=reducer)
        grads[name + ".weight"], grads[name + ".bias"] = gg, gbeta
        h_in, w_in = (saved.x.shape[2], saved.x.shape[3]) if s == 0 else saved.dims[s - 1]
        shp5 = ConvShape(B, h_in, w_in, cin, c, 5, 2)
        if s == 0:
            grads["model.conv1.weight"] = lane.run(
                lambda gz=gz: _wgrad_c1(eng, shp5, saved.x, gz, buckets.views["model.conv1.weight"], inv), gz)
        else:
            x_in = saved.acts[f"stage{s}.c"]
            grads[f"model.conv{i}.weight"] = lane.run(
                lambda gz=gz, x_in=x_in: _wgrad(eng, shp5, x_in, gz, buckets.views[f"model.conv{i}.weight"], inv), gz)
            # dL/d(stage-below output): ONE 3x3 convolution over dL/dz whose 4 cin output channels are the parity classes
            # of the stride-2 data gradient; the layer below reads that layout in place and masks it itself
            g = eng.conv_f16(gz, sw.conv_dgrad_f16, B, h, w, c, 4 * cin, 3, 1)
            g_parity, g_masked = True, False
        lane.run(lambda: buckets.done(s))       # this stage's three filter gradients are enqueued: reduce them now
    lane.join()
    buckets.finish()
    if overflow_flag is not None:
        # Loss scaling is STATIC (`loss_scale`), and ds_scale_cast_f32_to_f16 does not saturate: a scaled gradient beyond
        # fp16's 65504 is an inf in the gradient tensors, then NaN in the BatchNorm-backward sums and in every filter
        # gradient downstream.  Any such value reaches the filter gradients of its own layer (the sums feed dL/dz, which
        # the filter gradient contracts), so one pass over the f32 gradient buckets (46 MB: ~10 us each at HBM rate) sees
        # it; under data parallelism the buckets are already the global sums, so every rank raises the same flag.  The
        # flag stays on the device: the fused optimizers read it there (optim._FusedBase.skip_flag).
        for flat in buckets.flat.values():
            lib.call("ds_nonfinite_flag_f32", eng._p(flat), flat.numel(), eng._p(overflow_flag), st)
    return grads
