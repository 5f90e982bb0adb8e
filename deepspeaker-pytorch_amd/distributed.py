"""Data-parallel training of the triplet network: one process per GPU, torch.distributed over RCCL/xGMI.

The reference has no multi-GPU code at all; this is the new capability named in
BASELINE.json's north_star.  Partitioning: rank r owns rows [r*B_loc, (r+1)*B_loc) of the
anchor / positive / negative batches.  Exchange steps per iteration:

1. BatchNorm batch statistics -- forward {sum x, sum x^2} and backward {sum dy, sum dy*xhat}: one tiny
   float64 all-reduce per BatchNorm LAYER and direction, carrying the sums of all three members (anchor /
   positive / negative forwards run in lock-step, Engine.forward_train_group): 24 per step.  With them an
   N-rank step reproduces the single-process step on the global batch.
2. Embeddings (+ speaker labels): RCCL all-gather so every rank sees the global batch for cross-GPU
   semi-hard negative mining; the gather is differentiable (its adjoint is a reduce-scatter), so the
   gradient of a negative mined on another GPU returns to the rank that owns it.
3. Filter / weight gradients: one flat bucket per stage (+ fc).  The gradient kernels write straight into
   their bucket and the bucket's all-reduce is launched from inside the backward pass as soon as the stage's
   last gradient kernel is enqueued (backward._GradBuckets), so it runs over xGMI while the earlier stages'
   backward kernels execute.  BatchNorm affine gradients need no exchange: they come from global sums.

xGMI is point-to-point (7 links per GPU); the only bandwidth-relevant message is the 46 MB of gradients,
issued as 5 bucket-sized collectives (stage 4 alone is 26 MB and goes first, with the most backward work
left to hide behind).
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


class _Handles:
    """wait() on several collective handles in issue order (what an rs_ag gradient exchange returns)"""

    def __init__(self, hs, after=None):
        self.hs, self.after = [h for h in hs if h is not None], after

    def wait(self):
        for h in self.hs:
            h.wait()
        if self.after is not None:
            self.after()
            self.after = None


class Reducer:
    """Thin handle on a process group (backend "nccl" = RCCL on ROCm; "gloo" in the CPU logic tests).

    Gradient exchange knobs (constructor arguments, else the environment, else the safe default):

    * `grad_comm` / DS_GRAD_COMM = "shared" (default) | "separate".  A process group runs ALL its collectives in issue
      order on one internal stream: with one communicator, a BatchNorm statistics all-reduce (main stream, critical
      path) issued after a gradient-bucket all-reduce (filter-gradient stream) waits for that stream's kernels --
      measured with one rank: 0 ms of overlap, +2.2 ms per step.  "separate" puts the buckets on a communicator of
      their own (`dist.new_group`, or a pre-built `grad_group`) and reduces each bucket from inside the backward pass.
      Two communicators driven concurrently from two streams of one device are only safe if every rank's device-side
      launch order agrees; that has never run on more than one real GPU here, so it is OPT-IN.  "shared" keeps one
      communicator and issues the buckets after the backward pass's last BatchNorm collective (no overlap with the
      pass; one program-ordered sequence of collectives per rank -- cannot deadlock).
    * `grad_reduce` / DS_GRAD_REDUCE = "allreduce" (default) | "rs_ag": the bucket exchange as ONE all-reduce
      (whatever algorithm RCCL picks) or as an explicit reduce-scatter + all-gather pair (the direct
      exchange that keeps all 7 xGMI links busy with 1/N-sized shards).  Same sums; A/B-able the day a node exists.

    `dist.new_group` is a collective over the DEFAULT group: with grad_comm="separate" and no `grad_group`, every
    rank of the job must construct its Reducer at the same point (enable_data_parallel), also ranks outside `group`.
    """

    def __init__(self, group: Optional[dist.ProcessGroup] = None, force: bool = False,
                 grad_comm: Optional[str] = None, grad_reduce: Optional[str] = None,
                 grad_group: Optional[dist.ProcessGroup] = None):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised: launch one process per GPU with "
                               "torch.distributed.run and call dist.init_process_group('nccl') first")
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        # `active`: take the data-parallel launch sequence (float64 sums -> all-reduce -> *_from_sums kernels, gradient
        # buckets reduced by the backward pass, all-gather / reduce-scatter of embeddings).  A group of one
        # has nothing to exchange and skips it -- unless `force` (or DS_FORCE_COLLECTIVES=1) asks for the very same
        # sequence an N-rank job runs, collectives included: how the path is exercised on a single GPU.
        self.active = self.world > 1 or force or os.environ.get("DS_FORCE_COLLECTIVES", "0") == "1"
        self.n_all_reduce = 0           # gradient / statistics exchanges issued so far (tests assert the per-step count)
        self.grad_comm = grad_comm or os.environ.get("DS_GRAD_COMM", "shared")
        self.grad_reduce = grad_reduce or os.environ.get("DS_GRAD_REDUCE", "allreduce")
        if self.grad_comm not in ("shared", "separate"):
            raise ValueError(f"grad_comm / DS_GRAD_COMM must be 'shared' or 'separate', got {self.grad_comm!r}")
        if self.grad_reduce not in ("allreduce", "rs_ag"):
            raise ValueError(f"grad_reduce / DS_GRAD_REDUCE must be 'allreduce' or 'rs_ag', got {self.grad_reduce!r}")
        self.grad_group = group
        if grad_group is not None:
            self.grad_group, self.grad_comm = grad_group, "separate"
        elif self.active and self.grad_comm == "separate":
            ranks = dist.get_process_group_ranks(group) if group is not None else None
            self.grad_group = dist.new_group(ranks=ranks)
        # buckets are reduced from inside the backward pass (overlapped) only when they have their own communicator
        self.overlap_gradients = self.grad_comm == "separate"

    def all_reduce_sum_(self, t: torch.Tensor, async_op: bool = False, gradients: bool = False):
        self.n_all_reduce += 1
        if gradients and self.grad_reduce == "rs_ag" and self.world > 1:
            return self._rs_ag_sum_(t, async_op)
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.grad_group if gradients else self.group,
                               async_op=async_op)

    def _rs_ag_sum_(self, flat: torch.Tensor, async_op: bool):
        """In-place sum over the ranks of a flat bucket as reduce-scatter + all-gather (each rank reduces one 1/N
        shard, then the shards are gathered back).  The bucket is padded to a multiple of the world size in a
        staging buffer when needed."""
        g, w = self.grad_group, self.world
        n = flat.numel()
        per = -(-n // w)
        src = flat.reshape(-1)
        staged = None
        if per * w != n or not src.is_contiguous():
            staged = torch.zeros(per * w, dtype=flat.dtype, device=flat.device)
            staged[:n].copy_(src)
            src = staged
        shard = torch.empty(per, dtype=flat.dtype, device=flat.device)
        if dist.get_backend(g) == "gloo":               # gloo (CPU logic tests) has no reduce-scatter
            full = src.clone()
            dist.all_reduce(full, op=dist.ReduceOp.SUM, group=g)
            shard.copy_(full[self.rank * per:(self.rank + 1) * per])
            h1 = None
        else:
            h1 = dist.reduce_scatter_tensor(shard, src, op=dist.ReduceOp.SUM, group=g, async_op=async_op)
        h2 = dist.all_gather_into_tensor(src, shard, group=g, async_op=async_op)
        after = None
        if staged is not None:
            def after():
                flat.reshape(-1).copy_(staged[:n])
        if async_op:
            return _Handles([h1, h2], after)
        if after is not None:
            after()
        return None

    def all_gather_rows(self, t: torch.Tensor) -> torch.Tensor:
        """[n, ...] on every rank -> [world*n, ...], rank-major (equal n on all ranks)."""
        out = torch.empty((self.world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out, t.contiguous(), group=self.group)
        return out

    def reduce_scatter_rows(self, t: torch.Tensor) -> torch.Tensor:
        """[world*n, ...] on every rank -> this rank's [n, ...] slice of the sum over ranks."""
        n = t.shape[0] // self.world
        if dist.get_backend(self.group) == "gloo":       # gloo (CPU logic tests) has no reduce-scatter
            full = t.contiguous().clone()
            dist.all_reduce(full, op=dist.ReduceOp.SUM, group=self.group)
            return full[self.rank * n:(self.rank + 1) * n].contiguous()
        out = torch.empty((n,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        dist.reduce_scatter_tensor(out, t.contiguous(), op=dist.ReduceOp.SUM, group=self.group)
        return out


class AllGatherRows(torch.autograd.Function):
    """Differentiable all-gather of embedding rows: backward = reduce-scatter of the gathered gradient."""

    @staticmethod
    def forward(ctx, x: torch.Tensor, reducer: Reducer):
        ctx.reducer = reducer
        return reducer.all_gather_rows(x)

    @staticmethod
    def backward(ctx, g):
        return ctx.reducer.reduce_scatter_rows(g), None


# names of gradients that must be summed over ranks (everything except BatchNorm affine gradients,
# which come out of the all-reduced statistics and are already global)
def needs_allreduce(name: str) -> bool:
    return not (".bn" in name)


def allreduce_gradients(grads: Dict[str, torch.Tensor], reducer: Reducer, n_buckets: int = 5) -> None:
    """Sum filter / fc gradients over the ranks in place: a few flat buckets (one per stage + fc),
    launched asynchronously back to back, then copied back.  Deterministic bucket composition."""
    if not reducer.active:
        return
    names = sorted(n for n in grads if needs_allreduce(n))
    buckets: List[List[str]] = [[] for _ in range(n_buckets)]
    for n in names:                                   # stage index from the name; fc -> last bucket
        k = n_buckets - 1
        for i in range(1, 5):
            if f"conv{i}." in n and "layer" not in n or f"layer{i}." in n:
                k = min(i - 1, n_buckets - 1)
        buckets[k].append(n)
    work = []
    for b in buckets:
        if not b:
            continue
        flat = torch.cat([grads[n].reshape(-1) for n in b])
        work.append((b, flat, reducer.all_reduce_sum_(flat, async_op=True, gradients=True)))
    for b, flat, h in work:
        h.wait()
        off = 0
        for n in b:
            k = grads[n].numel()
            grads[n].copy_(flat[off:off + k].view_as(grads[n]))
            off += k


@dataclass
class TripletStepResult:
    loss: torch.Tensor                    # global-batch mean hinge (identical on every rank)
    grads: Dict[str, torch.Tensor]        # global parameter gradients (identical on every rank)
    embeddings: Tuple[torch.Tensor, torch.Tensor, torch.Tensor]
    mined: Optional[torch.Tensor] = None  # index into the gathered candidate set per local anchor


def triplet_train_step(eng, pw, bns, bn_weights, xa, xp, xn, margin: float, reducer: Optional[Reducer] = None,
                       labels: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
                       mine: bool = False, arith: str = "f32", loss_scale: float = 1024.0) -> TripletStepResult:
    """One data-parallel training step on this rank's shard of the triplet batch (engine level).

    reference semantics (train_triplet.py:215-224): three train-mode forwards (three BatchNorm statistic
    sets), TripletMarginLoss over the batch, backward.  With `reducer` the statistics, the loss mean and
    the gradients are those of the GLOBAL batch.  With `mine` (needs `labels` = (c1, c2) speaker ids of
    positives / negatives) every anchor's negative is replaced by the semi-hard negative found among the
    all-gathered embeddings of all ranks before the loss is taken.  `arith` "f16": the opt-in fp16 step (train_f16.py;
    `pw` packed with with_f16=True, with_f16_dgrad=True), same exchange pattern.
    """
    from .backward import backward_train
    lib = eng.lib
    world = reducer.world if reducer is not None else 1
    # the three forwards in lock-step over one batch: one statistics all-reduce per BatchNorm layer
    if arith == "f16":
        from .train_f16 import backward_train_f16, forward_train_group_f16
        (ea, ep, en), saved = forward_train_group_f16(eng, [xa, xp, xn], pw, bns, save=True, reducer=reducer)
    else:
        (ea, ep, en), saved = eng.forward_train_group([xa, xp, xn], pw, bns, save=True, reducer=reducer)
    ea, ep, en = ea.contiguous(), ep.contiguous(), en.contiguous()
    n_loc, d = ea.shape
    n_glob = n_loc * world
    st = eng._stream(ea)
    mined = None
    gathered = None
    if mine:
        if labels is None:
            raise ValueError("mining needs the speaker labels (c1, c2) of the batch")
        c1, c2 = labels
        loc = torch.cat([ea, ep, en])
        lab = torch.cat([c1, c1, c2]).to(torch.int64)
        if reducer is not None and reducer.active:
            # [world][3*n_loc] rank-major candidate set and its labels
            gathered = reducer.all_gather_rows(loc)
            glab = reducer.all_gather_rows(lab)
        else:
            gathered, glab = loc, lab
        d_p = eng.pairwise_distance(ea, ep)
        mined = torch.empty(n_loc, dtype=torch.int64, device=ea.device)
        mws = torch.empty(lib.raw("ds_mine_workspace_floats")(n_loc, gathered.shape[0]), dtype=torch.float32,
                          device=ea.device)
        lib.call("ds_mine_semihard_f32", eng._p(ea), eng._p(d_p), eng._p(c1.to(torch.int64).contiguous()),
                 eng._p(gathered), eng._p(glab), eng._p(mws), eng._p(mined), None, n_loc, gathered.shape[0], d, st)
        en_used = torch.empty_like(en)
        lib.call("ds_gather_rows_f32", eng._p(gathered), eng._p(mined), eng._p(en_used), n_loc, d, st)
    else:
        en_used = en
    loss_loc, d_p, d_n = eng.triplet_margin(ea, ep, en_used, margin)
    # global mean: local mean * n_loc / n_glob, summed over ranks
    loss = loss_loc * (float(n_loc) / float(n_glob))
    if reducer is not None and reducer.active:
        reducer.all_reduce_sum_(loss)
    gl = torch.full((1,), float(n_loc) / float(n_glob), dtype=torch.float32, device=ea.device)
    ga, gp, gn_used = torch.empty_like(ea), torch.empty_like(ep), torch.empty_like(en)
    lib.call("ds_triplet_margin_bwd_f32", eng._p(ea), eng._p(ep), eng._p(en_used), eng._p(d_p), eng._p(d_n),
             float(margin), eng._p(gl), eng._p(ga), eng._p(gp), eng._p(gn_used), n_loc, d, st)
    if mine:
        # adjoint of the row gather, then of the all-gather: gradients of mined candidates go home
        gcand = torch.empty_like(gathered)
        lib.call("ds_scatter_add_rows_f32", eng._p(gn_used), eng._p(mined), eng._p(gcand), n_loc,
                 gathered.shape[0], d, 0, st)
        if reducer is not None and reducer.active:
            gcand = reducer.reduce_scatter_rows(gcand)
        ga = ga + gcand[:n_loc]                         # elementwise accumulation of two gradient paths
        gp = gp + gcand[n_loc:2 * n_loc]
        gn = gcand[2 * n_loc:].contiguous()
    else:
        gn = gn_used
    # one backward pass over the concatenated batch; its per-stage gradient buckets are all-reduced as they
    # are produced (backward._GradBuckets), overlapped with the earlier stages' kernels
    if arith == "f16":
        grads = backward_train_f16(eng, bn_weights, pw, saved, torch.cat([ga, gp, gn]).contiguous(), loss_scale=loss_scale,
                                   reducer=reducer, reduce_gradients=reducer is not None)
    else:
        grads = backward_train(eng, bn_weights, pw, saved, torch.cat([ga, gp, gn]).contiguous(), reducer=reducer,
                               reduce_gradients=reducer is not None)
    return TripletStepResult(loss.reshape(()), grads, (ea, ep, en), mined)
