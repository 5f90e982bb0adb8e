"""One whole triplet training step as ONE HIP graph (round 6; the round-5 review's item 5).

The reference's step (train_triplet.py:215-224) -- `model(data_a), model(data_p), model(data_n)`, `TripletMarginLoss`,
`loss.backward()`, `optimizer.step()` -- is ~330 (fp16) / ~400 (f32-class) kernel launches over two or four HIP streams
here.  `GraphedTripletStep` captures all of it once and replays it per batch:

    step = GraphedTripletStep(model, optimizer, margin=0.1, example=(xa, xp, xn))
    for xa, xp, xn in batches:
        loss = step(xa, xp, xn)            # device tensor, valid until the next call; no host work besides the replay

What makes the step capturable (each was a capture error or a silent wrong answer before):
  * the library allocates nothing and never synchronises; the persistent kernels' tile counters come from the caller's
    workspace (`_native.NativeLib.add_sched_workspace`), one slot per captured launch for good;
  * the fused optimizers write their pointer tables with `ds_fill_bytes` (values as kernel arguments) instead of a pinned
    staging copy, whose event torch's host allocator may not query once it was recorded in a capturing stream;
  * the step count that enters Adagrad's decayed learning rate / Adam's bias corrections lives ON THE DEVICE
    (`optimizer.enable_device_step()`): a replay uses the count of that replay, not of the capture;
  * the fp16 step's overflow flag is raised, consumed and cleared on the device inside the graph.
Weights change every replay, so the filter re-pack at the head of the step is part of the graph.  Host-side bookkeeping
that a replay cannot do -- autograd version counters, `state[p]["step"]`, `num_batches_tracked` is a device tensor and
IS updated -- is brought up to date by `sync()` (called automatically before the object hands the model back:
`close()` / context exit), which is one host synchronisation.

Frozen at capture (kernel arguments of the graph's nodes): the batch shape, the margin, the optimizer's hyper-parameters
(learning rate, momentum, betas, weight decay -- Adagrad's DECAY of the learning rate is not frozen: it is computed from the
device-side step count), the loss scale of the fp16 step.  Change any of them -> build a new GraphedTripletStep.

The warm-up the capture needs (allocator pools, launch plans, LDS opt-ins) runs REAL steps; parameters, BatchNorm
buffers, optimizer state and the loss-scale flag are restored in place afterwards, so constructing the object leaves the
model where it was.  Measured on MI355X (tools/train_graph_probe.py): the replayed fp16 step reproduces the eager loss
trajectory digit for digit and takes 9.2 ms against 8.7 ms eager (the graph's own stream assignment overlaps the
filter-gradient branch less well) -- what it buys is a step with no host work and regions that agree within 1.5 %.
"""
from __future__ import annotations

from typing import Sequence

import torch

from .model import DeepSpeakerModel, TripletMarginLoss, _require_cuda


class GraphedTripletStep:
    def __init__(self, model: DeepSpeakerModel, optimizer, margin: float = 0.1, example: Sequence[torch.Tensor] = (),
                 warmup: int = 3):
        if len(example) != 3:
            raise ValueError("example: the three [B,1,T,64] batches (anchor, positive, negative) of one step")
        for t in example:
            _require_cuda(t, "GraphedTripletStep")
        if not model.training:
            raise RuntimeError("GraphedTripletStep captures the TRAINING step: call model.train() first")
        red = getattr(model, "_reducer", None)
        if red is not None and getattr(red, "active", False):
            raise RuntimeError("GraphedTripletStep: a data-parallel step has collectives on its critical path (BatchNorm "
                               "statistics, gradient buckets); capturing those is not supported -- use the eager step")
        self.model, self.optimizer = model, optimizer
        self.loss_fn = TripletMarginLoss(margin)
        self.static_x = [t.detach().contiguous().float().clone() for t in example]
        dev = self.static_x[0].device
        if hasattr(optimizer, "enable_device_step"):
            optimizer.enable_device_step()
            optimizer._device_step([p for g in optimizer.param_groups for p in g["params"]])    # exists before the warm-up
        # ---- warm-up with real steps, then everything they touched is put back IN PLACE (pointers must not change) ----
        keep_p = [p.detach().clone() for p in model.parameters()]
        keep_b = [b.detach().clone() for b in model.buffers()]
        keep_s = {p: {k: (v.clone() if torch.is_tensor(v) else v) for k, v in optimizer.state.get(p, {}).items()}
                  for g in optimizer.param_groups for p in g["params"]}
        keep_dev_step = optimizer._dev_step.clone() if getattr(optimizer, "_dev_step", None) is not None else None
        keep_updates = model._stat_updates
        cur = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                self._one_step()
        cur.wait_stream(side)
        torch.cuda.synchronize(dev)
        with torch.no_grad():
            for p, k in zip(model.parameters(), keep_p):
                p.copy_(k)
            for b, k in zip(model.buffers(), keep_b):
                b.copy_(k)
            for p, st in keep_s.items():
                cur_st = optimizer.state.get(p, {})
                for name in list(cur_st):
                    if name not in st:              # state the warm-up created: zero it (as a first step would find it)
                        if torch.is_tensor(cur_st[name]):
                            cur_st[name].zero_()
                    elif torch.is_tensor(cur_st[name]):
                        cur_st[name].copy_(st[name])
                    else:
                        cur_st[name] = st[name]
            if keep_dev_step is not None:
                optimizer._dev_step.copy_(keep_dev_step)
            flag = model.__dict__.get("_overflow_state")
            if flag is not None:
                flag.zero_()
        model._stat_updates = keep_updates
        torch.cuda.synchronize(dev)
        # ---- capture ----
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.static_loss, self.static_out = self._one_step()
        self.replays = 0

    def _one_step(self):
        out = self.model.forward_triplet(*self.static_x)
        loss = self.loss_fn.forward(*out)
        self.optimizer.zero_grad(set_to_none=True)
        loss.backward()
        self.optimizer.step()
        return loss.detach(), tuple(o.detach() for o in out)

    def __call__(self, data_a: torch.Tensor, data_p: torch.Tensor, data_n: torch.Tensor) -> torch.Tensor:
        for dst, src in zip(self.static_x, (data_a, data_p, data_n)):
            if tuple(src.shape) != tuple(dst.shape):
                raise ValueError(f"captured for batches of shape {tuple(dst.shape)}, got {tuple(src.shape)}")
            dst.copy_(src)
        self.graph.replay()
        self.replays += 1
        return self.static_loss

    @property
    def embeddings(self):
        """(out_a, out_p, out_n) of the last replayed step (static tensors: valid until the next call)."""
        return self.static_out

    def sync(self):
        """Bring the host-side bookkeeping the replays could not do up to date: parameter version counters (the packed-
        filter / folded-BatchNorm caches of eval forwards key on them), `state[p]["step"]`, the model's BatchNorm update
        count.  One host synchronisation (the device-side step count is read)."""
        if self.replays == 0:
            return
        for g in self.optimizer.param_groups:
            for p in g["params"]:
                torch.autograd.graph.increment_version(p)
        if hasattr(self.optimizer, "sync_host_steps"):
            self.optimizer.sync_host_steps()
        self.model._stat_updates += 3 * self.replays
        self.replays = 0

    def close(self):
        self.sync()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False
