"""Backward pass of the train-mode ResCNN as a sequence of C-ABI launches.

Restates what torch autograd derives for `loss.backward()` (reference train_triplet.py:223,290) over
`DeepSpeakerModel.forward` (model.py:185-218): l2-norm, fc, temporal mean, and per stage
clip / BatchNorm(train) / 3x3 conv x2 with the identity residual, then clip / BatchNorm / 5x5 s2 conv.
Every op is a HIP kernel (include/deepspeaker_hip.h, "backward" sections); this file only orders them.
"""
from __future__ import annotations

import ctypes
from typing import Dict


import torch

from ._native import ConvShape
from .engine import ALPHA, L2_EPS, STAGE_CHANNELS, Engine, PackedWeights, SavedForward


def _bn_bwd_group(eng: Engine, g1, g2, act, z, stats, gamma, reducer=None):
    """_bn_bwd over a batch made of len(stats) members with their own batch statistics (the three forwards of a
    triplet step run as one batch, Engine.forward_train_group): the reductions run per member on slices of the
    batch; with a `reducer` the sums of ALL members travel in one all-reduce.  dgamma / dbeta are summed over the
    members (the reference accumulates them over its three backward passes)."""
    G = len(stats)
    c = z.shape[-1]
    Bm = z.shape[0] // G
    dev = z.device
    n_pix = (z.numel() // c) // G
    rows = eng.lib.raw("ds_bn_bwd_partial_rows")(n_pix, c)
    gy, gz = torch.empty_like(z), torch.empty_like(z)
    member_sums = torch.empty((2, G, c), dtype=torch.float32, device=dev)      # dgamma / dbeta per member
    gg_all, gb_all = member_sums[0], member_sums[1]
    st = eng._stream(z)

    def m(t, g):
        return None if t is None else t[g * Bm:(g + 1) * Bm]

    dp = reducer is not None and reducer.active
    partial = torch.empty((G, rows, c, 2), dtype=torch.float32, device=dev)
    coef = torch.empty((G, 3 * c), dtype=torch.float32, device=dev)
    # the members' statistics as consecutive rows of one table (Engine.forward_train_group lays them out so): all
    # members' reductions, coefficient sets and applications in four launches instead of 3 G + 2
    step = c * 4
    tabled = all(stats[g][0].data_ptr() == stats[0][0].data_ptr() + g * step
                 and stats[g][1].data_ptr() == stats[0][1].data_ptr() + g * step for g in range(G))
    if not dp and tabled:
        gg, gb = torch.empty(c, dtype=torch.float32, device=dev), torch.empty(c, dtype=torch.float32, device=dev)
        eng.lib.call("ds_bn_bwd_group_f32", eng._p(g1), eng._p(g2), eng._p(act), eng._p(z), eng._p(stats[0][0]),
                     eng._p(stats[0][1]), eng._p(gamma.detach()), eng._p(gy), eng._p(partial), eng._p(coef),
                     eng._p(member_sums), eng._p(gg), eng._p(gb), eng._p(gz), n_pix, c, G, st)
        return gy, gz, gg, gb
    if dp and tabled:
        # the same grouped launches, split where the sums of ALL members travel in one all-reduce (2 + 3 launches)
        sums = torch.empty((G, 2 * c + 1), dtype=torch.float64, device=dev)
        eng.lib.call("ds_bn_bwd_group_reduce_f32", eng._p(g1), eng._p(g2), eng._p(act), eng._p(z), eng._p(stats[0][0]),
                     eng._p(stats[0][1]), eng._p(gy), eng._p(partial), eng._p(sums), n_pix, c, G, st)
        reducer.all_reduce_sum_(sums)
        gg, gb = torch.empty(c, dtype=torch.float32, device=dev), torch.empty(c, dtype=torch.float32, device=dev)
        eng.lib.call("ds_bn_bwd_group_apply_f32", eng._p(sums), eng._p(gy), eng._p(z), eng._p(stats[0][0]),
                     eng._p(stats[0][1]), eng._p(gamma.detach()), eng._p(coef), eng._p(member_sums), eng._p(gg), eng._p(gb),
                     eng._p(gz), n_pix, c, G, st)
        return gy, gz, gg, gb
    if not dp:
        for g in range(G):
            mean, invstd = stats[g][0], stats[g][1]
            eng.lib.call("ds_bn_bwd_f32", eng._p(m(g1, g)), eng._p(m(g2, g)), eng._p(m(act, g)), eng._p(m(z, g)),
                         eng._p(mean), eng._p(invstd), eng._p(gamma.detach()), eng._p(m(gy, g)), eng._p(partial[g]),
                         eng._p(coef[g]), eng._p(gg_all[g]), eng._p(gb_all[g]), eng._p(m(gz, g)), n_pix, c, st)
    else:
        sums = torch.empty((G, 2 * c + 1), dtype=torch.float64, device=dev)
        for g in range(G):
            mean, invstd = stats[g][0], stats[g][1]
            eng.lib.call("ds_bn_bwd_reduce_f32", eng._p(m(g1, g)), eng._p(m(g2, g)), eng._p(m(act, g)), eng._p(m(z, g)),
                         eng._p(mean), eng._p(invstd), eng._p(m(gy, g)), eng._p(partial[g]), n_pix, c, st)
            eng.lib.call("ds_partial_sum_f64", eng._p(partial[g]), rows, eng._p(sums[g]), c, st)
        sums[:, 2 * c] = float(n_pix)
        reducer.all_reduce_sum_(sums)                               # every member of this layer in ONE collective
        for g in range(G):
            mean, invstd = stats[g][0], stats[g][1]
            eng.lib.call("ds_bn_bwd_apply_f32", eng._p(sums[g]), 0, eng._p(m(gy, g)), eng._p(m(z, g)), eng._p(mean),
                         eng._p(invstd), eng._p(gamma.detach()), eng._p(coef[g]), eng._p(gg_all[g]), eng._p(gb_all[g]),
                         eng._p(m(gz, g)), n_pix, c, st)
    gg, gb = torch.empty(c, dtype=torch.float32, device=dev), torch.empty(c, dtype=torch.float32, device=dev)
    eng.lib.call("ds_colsum_f32", eng._p(gg_all), eng._p(gg), G, c, st)
    eng.lib.call("ds_colsum_f32", eng._p(gb_all), eng._p(gb), G, c, st)
    return gy, gz, gg, gb


def _bn_bwd(eng: Engine, g1, g2, act, z, stats, gamma, reducer=None):
    """(masked upstream gradient gy, gz = dL/d(conv output), dgamma, dbeta).  With a `reducer` the two
    per-channel sums are all-reduced between the reduce and the apply kernel (global-batch BatchNorm);
    dgamma / dbeta then already are the global gradients."""
    if isinstance(stats, list):
        return _bn_bwd_group(eng, g1, g2, act, z, stats, gamma, reducer)
    mean, invstd = stats[0], stats[1]
    c = z.shape[-1]
    n_pix = z.numel() // c
    dev = z.device
    rows = eng.lib.raw("ds_bn_bwd_partial_rows")(n_pix, c)
    gy = torch.empty_like(z)
    gz = torch.empty_like(z)
    partial = torch.empty((rows, c, 2), dtype=torch.float32, device=dev)
    coef = torch.empty(3 * c, dtype=torch.float32, device=dev)
    gg = torch.empty(c, dtype=torch.float32, device=dev)
    gb = torch.empty_like(gg)
    if reducer is not None and reducer.active:
        st = eng._stream(z)
        eng.lib.call("ds_bn_bwd_reduce_f32", eng._p(g1), eng._p(g2), eng._p(act), eng._p(z), eng._p(mean),
                     eng._p(invstd), eng._p(gy), eng._p(partial), n_pix, c, st)
        sums = torch.empty(2 * c + 1, dtype=torch.float64, device=dev)
        eng.lib.call("ds_partial_sum_f64", eng._p(partial), rows, eng._p(sums), c, st)
        sums[2 * c] = float(n_pix)
        reducer.all_reduce_sum_(sums)
        eng.lib.call("ds_bn_bwd_apply_f32", eng._p(sums), 0, eng._p(gy), eng._p(z),
                     eng._p(mean), eng._p(invstd), eng._p(gamma.detach()), eng._p(coef), eng._p(gg), eng._p(gb),
                     eng._p(gz), n_pix, c, st)
        return gy, gz, gg, gb
    eng.lib.call("ds_bn_bwd_f32", eng._p(g1), eng._p(g2), eng._p(act), eng._p(z), eng._p(mean), eng._p(invstd),
                 eng._p(gamma.detach()), eng._p(gy), eng._p(partial), eng._p(coef), eng._p(gg), eng._p(gb),
                 eng._p(gz), n_pix, c, eng._stream(z))
    return gy, gz, gg, gb


FUSE_DGRAD_BN_BWD = True       # module switch for A/B runs and tests of the unfused sequence


def _dgrad_bn_bwd(eng: Engine, shp: ConvShape, gz_up, bank_bf16, g2, z, stats, gamma, reducer=None):
    """`_dgrad` of a 3x3 layer followed by `_bn_bwd` of the BatchNorm + clipped-ReLU layer it feeds, with the first half
    of the latter inside the data-gradient kernel's epilogue (ds_conv_dgrad_bnbwd_bf16): the gradient never makes the
    round trip through HBM between the two, and the mask comes from the layer's own pre-activation `z` instead of a
    third tensor.  Returns (gy, gz, dgamma, dbeta) like `_bn_bwd`, or None where the fused kernel does not apply (a tile
    of the launch would straddle two members, statistics not laid out as tables): the caller then runs the two steps."""
    if not FUSE_DGRAD_BN_BWD or bank_bf16 is None:
        return None
    members = stats if isinstance(stats, list) else [stats]
    G = len(members)
    c = z.shape[-1]
    step = c * 4
    if any(len(m_) < 4 for m_ in members):
        return None
    if not all(members[g][k].data_ptr() == members[0][k].data_ptr() + g * step for g in range(G) for k in range(4)):
        return None
    rows = eng.lib.raw("ds_conv_dgrad_bnbwd_bf16_rows")(ctypes.byref(shp), G)
    if rows <= 0:
        return None
    dev = z.device
    n_pix = (z.numel() // c) // G
    st = eng._stream(z)
    mean, invstd, sc, sh = members[0]
    gy, gz = torch.empty_like(z), torch.empty_like(z)
    partial = torch.empty((G, rows, c, 2), dtype=torch.float32, device=dev)
    coef = torch.empty((G, 3 * c), dtype=torch.float32, device=dev)
    member_sums = torch.empty((2, G, c), dtype=torch.float32, device=dev)
    gg, gb = torch.empty(c, dtype=torch.float32, device=dev), torch.empty(c, dtype=torch.float32, device=dev)
    eng.lib.call("ds_conv_dgrad_bnbwd_bf16", ctypes.byref(shp), eng._p(gz_up), eng._p(bank_bf16[0]), eng._p(bank_bf16[1]),
                 eng._p(g2), eng._p(z), eng._p(mean), eng._p(invstd), eng._p(sc), eng._p(sh), G, eng._p(gy),
                 eng._p(partial), st)
    if reducer is not None and reducer.active:
        sums = torch.empty((G, 2 * c + 1), dtype=torch.float64, device=dev)
        eng.lib.call("ds_partial_sum_f64_group", eng._p(partial), rows, eng._p(sums), n_pix, c, G, st)
        reducer.all_reduce_sum_(sums)                               # every member of this layer in ONE collective
        eng.lib.call("ds_bn_bwd_group_apply_f32", eng._p(sums), eng._p(gy), eng._p(z), eng._p(mean), eng._p(invstd),
                     eng._p(gamma.detach()), eng._p(coef), eng._p(member_sums), eng._p(gg), eng._p(gb), eng._p(gz),
                     n_pix, c, G, st)
    else:
        eng.lib.call("ds_bn_bwd_group_finish_f32", eng._p(partial), rows, eng._p(gy), eng._p(z), eng._p(mean),
                     eng._p(invstd), eng._p(gamma.detach()), eng._p(coef), eng._p(member_sums), eng._p(gg), eng._p(gb),
                     eng._p(gz), n_pix, c, G, st)
    return gy, gz, gg, gb


def _dgrad_s2_bn_bwd(eng: Engine, shp: ConvShape, gz_up, bank_bf16, act, z, stats, gamma, reducer=None):
    """`_dgrad` of a 5x5 stride-2 layer followed by `_bn_bwd` of the BasicBlock output it feeds (bn2 + residual + clip of
    the previous stage), fused like `_dgrad_bn_bwd` (ds_conv_dgrad_s2_bnbwd_bf16): the four parity-class launches mask
    with the stored activation `act`, sum and write gy.  Returns (gy, gz, dgamma, dbeta) or None (not applicable)."""
    if not FUSE_DGRAD_BN_BWD or bank_bf16 is None:
        return None
    members = stats if isinstance(stats, list) else [stats]
    G = len(members)
    c = z.shape[-1]
    step = c * 4
    if not all(members[g][k].data_ptr() == members[0][k].data_ptr() + g * step for g in range(G) for k in range(2)):
        return None
    rows = eng.lib.raw("ds_conv_dgrad_s2_bnbwd_bf16_rows")(ctypes.byref(shp), G)
    if rows <= 0:
        return None
    dev = z.device
    n_pix = (z.numel() // c) // G
    st = eng._stream(z)
    mean, invstd = members[0][0], members[0][1]
    gy, gz = torch.empty_like(z), torch.empty_like(z)
    partial = torch.empty((G, rows, c, 2), dtype=torch.float32, device=dev)
    coef = torch.empty((G, 3 * c), dtype=torch.float32, device=dev)
    member_sums = torch.empty((2, G, c), dtype=torch.float32, device=dev)
    gg, gb = torch.empty(c, dtype=torch.float32, device=dev), torch.empty(c, dtype=torch.float32, device=dev)
    eng.lib.call("ds_conv_dgrad_s2_bnbwd_bf16", ctypes.byref(shp), eng._p(gz_up), eng._p(bank_bf16[0]),
                 eng._p(bank_bf16[1]), eng._p(act), eng._p(z), eng._p(mean), eng._p(invstd), G, eng._p(gy),
                 eng._p(partial), st)
    if reducer is not None and reducer.active:
        sums = torch.empty((G, 2 * c + 1), dtype=torch.float64, device=dev)
        eng.lib.call("ds_partial_sum_f64_group", eng._p(partial), rows, eng._p(sums), n_pix, c, G, st)
        reducer.all_reduce_sum_(sums)
        eng.lib.call("ds_bn_bwd_group_apply_f32", eng._p(sums), eng._p(gy), eng._p(z), eng._p(mean), eng._p(invstd),
                     eng._p(gamma.detach()), eng._p(coef), eng._p(member_sums), eng._p(gg), eng._p(gb), eng._p(gz),
                     n_pix, c, G, st)
    else:
        eng.lib.call("ds_bn_bwd_group_finish_f32", eng._p(partial), rows, eng._p(gy), eng._p(z), eng._p(mean),
                     eng._p(invstd), eng._p(gamma.detach()), eng._p(coef), eng._p(member_sums), eng._p(gg), eng._p(gb),
                     eng._p(gz), n_pix, c, G, st)
    return gy, gz, gg, gb


def _wgrad(eng: Engine, shp: ConvShape, x, gz, out_shape, fc_f: int = 0, x3: bool = False, out=None):
    """filter gradient into `out` (a contiguous view of a gradient bucket) or a fresh tensor"""
    if x3 and shp.KS in (3, 5) and shp.Cin % 64 == 0:      # split-operand bf16 matrix cores
        n_ws = eng.lib.raw("ds_conv_wgrad_bf16_workspace_floats")(ctypes.byref(shp))
        if n_ws <= 0:
            raise RuntimeError(f"ds_conv_wgrad_bf16_workspace_floats failed: {n_ws}")
        ws = torch.empty(n_ws, dtype=torch.float32, device=x.device)
        gw = out if out is not None else torch.empty(out_shape, dtype=torch.float32, device=x.device)
        eng.lib.call("ds_conv_wgrad_bf16", ctypes.byref(shp), eng._p(x), eng._p(gz), eng._p(ws), eng._p(gw),
                     eng._stream(x))
        return gw
    n_ws = eng.lib.raw("ds_conv_wgrad_workspace_floats")(ctypes.byref(shp))
    if n_ws <= 0:
        raise RuntimeError(f"ds_conv_wgrad_workspace_floats failed: {n_ws}")
    ws = torch.empty(n_ws, dtype=torch.float32, device=x.device)
    gw = out if out is not None else torch.empty(out_shape, dtype=torch.float32, device=x.device)
    eng.lib.call("ds_conv_wgrad_f32", ctypes.byref(shp), eng._p(x), eng._p(gz), eng._p(ws), eng._p(gw), fc_f,
                 eng._stream(x))
    return gw


def _dgrad(eng: Engine, shp: ConvShape, gz, w_dgrad, w_dgrad_bf16=None):
    gx = torch.empty((shp.B, shp.H, shp.W, shp.Cin), dtype=torch.float32, device=gz.device)
    if w_dgrad_bf16 is not None:       # bf16x3 data gradient (3x3 stride 1 / 5x5 stride 2)
        eng.lib.call("ds_conv_dgrad_bf16", ctypes.byref(shp), eng._p(gz), eng._p(w_dgrad_bf16[0]),
                     eng._p(w_dgrad_bf16[1]), eng._p(gx), eng._stream(gz))
        return gx
    eng.lib.call("ds_conv_dgrad_f32", ctypes.byref(shp), eng._p(gz), eng._p(w_dgrad), eng._p(gx), eng._stream(gz))
    return gx


_wgrad_streams: Dict[tuple, "torch.cuda.Stream"] = {}
OVERLAP_FILTER_GRADIENTS = True        # module default of backward_train(overlap_filter_gradients=None); tools flip it for A/B runs


class _FilterGradLane:
    """Where the filter-gradient kernels run.  A layer's filter gradient is a leaf of the backward pass: nothing
    downstream waits for it, while the data gradient -> BatchNorm-backward chain next to it is what the earlier
    layers wait for, and that chain is half HBM-bound element-wise passes.  On the GPU the filter gradients are
    therefore enqueued on a second HIP stream (fork after the kernel that produced dL/d(conv output), join at the end
    of the pass): the matrix-core-bound gradient kernels and the HBM-bound BatchNorm passes then share the chip
    instead of queueing behind each other.  Results are those of the one-stream order (same kernels, same inputs).
    On the host emulator (CPU tensors) everything stays in program order."""

    def __init__(self, device: torch.device, enabled: bool = True, priority: int = 0):
        """`priority` -1: a high-priority stream (tuning probe).  HIP deals normal-priority streams to its (4) hardware
        queues round-robin in creation order, so whether this lane shares a queue -- and then serialises -- with the
        caller's stream depends on how many streams the process made before (fp16 step: 8.8 ms in three alignments of
        four, 9.6 in the fourth; `bench.py --pad-streams`).  A high-priority lane lives on queues of its own and measured
        8.77 - 8.79 ms in all four alignments of a fresh process -- but 13.0 ms inside the full bench process (after the
        bf16x3 legs), and 23.5 ms for the bf16x3 step in one alignment: priority also reorders dispatch.  Default 0."""
        self.main = self.side = None
        self.keep = []                  # main-stream tensors the side stream reads: alive until the join
        if device.type == "cuda" and enabled:
            self.main = torch.cuda.current_stream(device)
            side = _wgrad_streams.get((device, priority))
            if side is None:
                side = _wgrad_streams[(device, priority)] = torch.cuda.Stream(device=device, priority=priority)
            self.side = side
            side.wait_stream(self.main)

    def run(self, fn, *inputs):
        """fn() on the side stream, ordered after everything enqueued on the main stream so far; `inputs` are the
        main-stream tensors it reads.  They are kept alive until join() -- after which the main stream is ordered
        behind the side stream, so their memory can be recycled the ordinary way.  (Tensor.record_stream instead
        made the caching allocator hold every such block back until the side stream had caught up: 54 GiB reserved
        for a 14 GiB working set.)"""
        if self.side is None:
            return fn()
        self.side.wait_event(self.main.record_event())
        self.keep.extend(t for t in inputs if t is not None)
        with torch.cuda.stream(self.side):
            return fn()

    def join(self):
        if self.side is not None:
            self.main.wait_stream(self.side)
        self.keep.clear()


class _GradBuckets:
    """The filter / fc gradients of one backward pass, laid out as one flat buffer per stage (+ one for fc): the
    gradient kernels write straight into views of their bucket.  Under data parallelism each bucket is summed over
    the ranks either the moment its last gradient kernel has been enqueued -- it then travels over xGMI while the
    earlier stages' backward kernels execute; needs the buckets' own communicator, Reducer(grad_comm="separate") --
    or, on the shared communicator (default), right after the pass.  BatchNorm affine gradients come out of the
    (already global) statistic sums and are not reduced."""

    def __init__(self, shapes: Dict[int, Dict[str, tuple]], device, reducer=None):
        self.reducer = reducer if (reducer is not None and reducer.active) else None
        self.views: Dict[str, torch.Tensor] = {}
        self.flat: Dict[int, torch.Tensor] = {}
        self.work = []
        self.deferred = []              # buckets whose exchange waits for finish() (Reducer.overlap_gradients False)
        for b, names in shapes.items():
            n = sum(int(torch.Size(shp).numel()) for shp in names.values())
            flat = torch.empty(n, dtype=torch.float32, device=device)
            self.flat[b] = flat
            off = 0
            for name, shp in names.items():
                k = int(torch.Size(shp).numel())
                self.views[name] = flat[off:off + k].view(shp)
                off += k

    def done(self, bucket: int):
        """the bucket's last gradient kernel has been enqueued (on the current stream)"""
        if self.reducer is None:
            return
        if self.reducer.overlap_gradients:      # own communicator: reduce now, under the earlier stages' kernels
            self.work.append(self.reducer.all_reduce_sum_(self.flat[bucket], async_op=True, gradients=True))
        else:                                   # shared communicator: after the pass's last BatchNorm collective
            self.deferred.append(bucket)

    def finish(self):
        """called on the main stream after the filter-gradient stream has joined it"""
        for b in self.deferred:
            self.work.append(self.reducer.all_reduce_sum_(self.flat[b], async_op=True, gradients=True))
        self.deferred = []
        for h in self.work:
            if h is not None:
                h.wait()
        self.work = []


def backward_train(eng: Engine, bn_weights: Dict[str, torch.Tensor], pw: PackedWeights, saved: SavedForward,
                   ge: torch.Tensor, reducer=None, precision: str = "f32",
                   reduce_gradients: bool = False, overlap_filter_gradients=None) -> Dict[str, torch.Tensor]:
    """Parameter gradients (reference key names, reference shapes) given dL/d(embedding) `ge` [B,512].
    precision "bf16x3": data and filter gradients of the 3x3 / 5x5 layers run on the bf16 matrix cores with
    split operands; conv1 and fc stay on the f32 matrix cores.  `reduce_gradients` (data parallelism): the
    per-stage gradient buckets are all-reduced over `reducer` as the pass produces them, overlapped with the rest
    of the pass; the returned gradients are then the global sums.  `overlap_filter_gradients`: see _FilterGradLane."""
    x3 = precision == "bf16x3"
    lane = _FilterGradLane(ge.device, OVERLAP_FILTER_GRADIENTS if overlap_filter_gradients is None else overlap_filter_gradients)
    lib = eng.lib
    grads: Dict[str, torch.Tensor] = {}
    n_stages = len(pw.stages)
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
    # ---- l2-norm x alpha (model.py:210-213) ----
    gf = torch.empty_like(f)
    lib.call("ds_l2norm_scale_bwd_f32", eng._p(f), eng._p(ge), eng._p(gf), B, n_out, ALPHA, L2_EPS, st)
    # ---- fc (model.py:209): bias, weight, input ----
    pooled = saved.pooled
    k = pooled.shape[1]
    gb = buckets.views["model.fc.bias"]
    lib.call("ds_colsum_f32", eng._p(gf), eng._p(gb), B, n_out, st)
    grads["model.fc.bias"] = gb
    c_last = STAGE_CHANNELS[n_stages - 1]
    f_bins = k // c_last
    grads["model.fc.weight"] = _wgrad(eng, ConvShape(1, B, 1, k, n_out, 1, 1), pooled, gf, (n_out, k), f_bins,
                                      out=buckets.views["model.fc.weight"])
    buckets.done(n_stages)
    ws = torch.empty(lib.raw("ds_fc_workspace_floats")(B, n_out, k), dtype=torch.float32, device=f.device)
    gpooled = torch.empty((B, k), dtype=torch.float32, device=f.device)
    lib.call("ds_fc_l2norm_fwd_f32", eng._p(gf), eng._p(pw.fc_dgrad), None, eng._p(ws), eng._p(gpooled), None, B,
             n_out, k, 1.0, 0.0, st)
    # ---- temporal mean + the clip of the last stage output (model.py:205-207) ----
    out = saved.acts[f"stage{n_stages}.c"]
    _, hr, wc, c = out.shape
    g = torch.empty_like(out)
    lib.call("ds_avgpool_time_bwd_f32", eng._p(gpooled), eng._p(out), eng._p(g), B, hr, wc, c, st)
    g_is_masked = True
    pre = None
    for s in reversed(range(n_stages)):
        i, c = s + 1, STAGE_CHANNELS[s]
        h, w = saved.dims[s]
        cin = 1 if s == 0 else STAGE_CHANNELS[s - 1]
        a_act, b_act, c_act = (saved.acts[f"stage{i}.{t}"] for t in "abc")
        shp3 = ConvShape(B, h, w, c, c, 3, 1)
        # out = clip(bn2(conv2(y)) + r)            (model.py:73-80)
        name = f"model.layer{i}.0.bn2"
        if pre is not None:       # the stage above already ran this step inside its 5x5 data gradient
            g_out, gz, gg, gbeta = pre
            pre = None
        else:
            g_out, gz, gg, gbeta = _bn_bwd(eng, g, None, None if g_is_masked else c_act, saved.raws[name],
                                           saved.stats[name], bn_weights[name], reducer)
        grads[name + ".weight"], grads[name + ".bias"] = gg, gbeta
        grads[f"model.layer{i}.0.conv2.weight"] = lane.run(
            lambda gz=gz: _wgrad(eng, shp3, b_act, gz, (c, c, 3, 3), x3=x3,
                                 out=buckets.views[f"model.layer{i}.0.conv2.weight"]), gz)
        # y = clip(bn1(conv1(r)))                  (model.py:69-71): conv2's data gradient + bn1's backward
        name = f"model.layer{i}.0.bn1"
        bank = pw.stages[s].l_conv2_dgrad_bf16 if x3 else None
        fused = _dgrad_bn_bwd(eng, shp3, gz, bank, None, saved.raws[name], saved.stats[name], bn_weights[name], reducer)
        if fused is None:
            g_y = _dgrad(eng, shp3, gz, pw.stages[s].l_conv2_dgrad, bank)
            fused = _bn_bwd(eng, g_y, None, b_act, saved.raws[name], saved.stats[name], bn_weights[name], reducer)
        _, gz, gg, gbeta = fused
        grads[name + ".weight"], grads[name + ".bias"] = gg, gbeta
        grads[f"model.layer{i}.0.conv1.weight"] = lane.run(
            lambda gz=gz: _wgrad(eng, shp3, a_act, gz, (c, c, 3, 3), x3=x3,
                                 out=buckets.views[f"model.layer{i}.0.conv1.weight"]), gz)
        # r = clip(bn_i(conv_i(x)));  dL/dr = conv-path + residual path   (model.py:187-189, 67, 79)
        name = f"model.bn{i}"
        bank = pw.stages[s].l_conv1_dgrad_bf16 if x3 else None
        fused = _dgrad_bn_bwd(eng, shp3, gz, bank, g_out, saved.raws[name], saved.stats[name], bn_weights[name], reducer)
        if fused is None:
            g_r = _dgrad(eng, shp3, gz, pw.stages[s].l_conv1_dgrad, bank)
            fused = _bn_bwd(eng, g_r, g_out, a_act, saved.raws[name], saved.stats[name], bn_weights[name], reducer)
        _, gz, gg, gbeta = fused
        grads[name + ".weight"], grads[name + ".bias"] = gg, gbeta
        h_in, w_in = (saved.x.shape[2], saved.x.shape[3]) if s == 0 else saved.dims[s - 1]
        shp5 = ConvShape(B, h_in, w_in, cin, c, 5, 2)
        x_in = saved.x if s == 0 else saved.acts[f"stage{s}.c"]
        grads[f"model.conv{i}.weight"] = lane.run(
            lambda gz=gz: _wgrad(eng, shp5, x_in, gz, (c, cin, 5, 5), x3=x3, out=buckets.views[f"model.conv{i}.weight"]), gz)
        lane.run(lambda: buckets.done(s))       # this stage's three filter gradients are enqueued: reduce them now
        if s > 0:
            below = f"model.layer{s}.0.bn2"                 # out = clip(bn2(conv2(y)) + r) of the stage below
            bank = pw.stages[s].conv_dgrad_bf16 if x3 else None
            pre = _dgrad_s2_bn_bwd(eng, shp5, gz, bank, x_in, saved.raws[below], saved.stats[below], bn_weights[below],
                                   reducer)
            if pre is None:
                g = _dgrad(eng, shp5, gz, pw.stages[s].conv_dgrad, bank)   # unmasked: the next bn2 step masks it
                g_is_masked = False
    lane.join()
    buckets.finish()
    return grads
