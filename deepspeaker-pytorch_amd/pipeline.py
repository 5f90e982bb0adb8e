"""Batches in flight: eval-mode embeddings of a sequence of batches with consecutive batches alternating over a few HIP
streams.

Why: a forward of the fp16 path is ~10 persistent matrix-core kernels with an HBM-bound first layer in front and five
small latency-bound launches (temporal mean, projection, norm, ...) behind; one batch at a time, every launch boundary
and that whole latency-bound end leave the matrix cores idle.  Consecutive batches are independent (eval-mode BatchNorm:
per-utterance results do not depend on the batch, reference model.py:185-213), so the next batch's kernels can fill those
holes: with two batches in flight the bench step (forward + loss + filter + search) runs at 1.82 - 1.90 ms instead of
2.02 - 2.12 (`pipelined` in the bench line, DESIGN_LOG.md section 5).  More than two buys nothing (3: -4 %, 4: +0 %).

Every stream has launch plans and activation buffers of its own (`Engine.forward_eval_planned` keys its plans by stream),
so batches in flight never share a buffer; results are the tensors `model(x)` returns, in order.
"""
from __future__ import annotations

from typing import Iterable, Iterator

import torch

from .model import _require_cuda


class BatchesInFlight:
    """`for e in BatchesInFlight(model)(batches): ...` -- embeddings of each batch, in order, valid on the stream that
    was current when the generator was created (the consumer's stream is ordered after each batch's forward before the
    batch's embeddings are handed out).  `batches`: any iterable of [B,1,T,64] device tensors; a batch must stay
    unmodified until its embeddings have been yielded."""

    def __init__(self, model, in_flight: int = 2):
        if in_flight < 1:
            raise ValueError("in_flight must be at least 1")
        self.model, self.in_flight = model, int(in_flight)
        self._streams = {}

    def _lanes(self, device):
        lanes = self._streams.get(device)
        if lanes is None:
            lanes = self._streams[device] = [torch.cuda.Stream(device=device) for _ in range(self.in_flight)]
        return lanes

    def __call__(self, batches: Iterable[torch.Tensor]) -> Iterator[torch.Tensor]:
        if self.model.training:
            raise RuntimeError("BatchesInFlight is an inference path: call model.eval() first")
        pending = []                        # (embeddings, event on their lane)
        consumer = None
        lanes = None

        def take():
            e, done = pending.pop(0)
            consumer.wait_event(done)
            e.record_stream(consumer)       # allocated on its lane, consumed here
            return e

        for i, x in enumerate(batches):
            _require_cuda(x, "BatchesInFlight")
            if consumer is None:
                consumer = torch.cuda.current_stream(x.device)
                lanes = self._lanes(x.device)
            lane = lanes[i % self.in_flight]
            # whatever the forward derives from the weights -- the arithmetic a precision-"f16" model's guard settles on
            # (it measures on new weights), the packed filters, the folded BatchNorm -- is built HERE, on the consumer's
            # stream, ahead of the lanes' wait: built inside one lane (first call, first eval after a training step or a
            # weight load), the next lane's forward would read it with no ordering after the kernels that write it
            prec = self.model.eval_precision(x.contiguous().float())
            self.model._packed(with_bf16=prec in ("bf16x3", "bf16"), with_f16=prec == "f16")
            self.model._folded()
            lane.wait_stream(consumer)      # the batch was produced on the consumer's stream
            with torch.cuda.stream(lane), torch.no_grad():
                x.record_stream(lane)
                e = self.model(x)
                pending.append((e, lane.record_event()))
            if len(pending) >= self.in_flight:
                yield take()
        while pending:
            yield take()
