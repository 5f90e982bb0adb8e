"""The fp16 eval path's precision guard: `DeepSpeakerModel(precision="f16")` stays inside north_star's 1e-3 by MEASURING.

fp16 operands carry 11 significand bits; what that does to the embeddings depends on the network.  Measured against the
reference (tests/test_gpu_offdist.py, tools/f16_error_budget.py): 3.6e-4 - 5.0e-4 on seeded weights whatever the inputs,
1.2e-5 on an Adagrad-trained network, but 0.86e-3 - 1.0e-3 on an SGD-trained one whose embeddings have spread apart -- AT
the contract.  The per-layer budget of that network (tools/f16_error_budget.py: every layer contributes 1.4e-4 - 3.9e-4,
filters and activations alike; exact residuals or error-diffused filter rounding move the total by < 10 %) says no layer
can be fixed to buy the margin back: the error is what 11 bits give on that network.  So the path measures itself and
escalates:

  * the first eval forward on a new generation of weights runs `rows` utterances TAKEN ACROSS THE BATCH (every
    B // rows-th row, starting at an offset that moves from check to check) through BOTH the fp16 path and the
    f32-class path (split-operand bf16, 1e-5 from the reference), reduces max |e16 - e32| / max |e32| on the device
    (`ds_max_abs_diff_f32`; a non-finite difference reads as an infinite error) and reads it back -- one host
    synchronisation per weight generation.  The rate limit (`min_gap` forwards between synchronous checks) applies
    only while the weights change on EVERY forward (an evaluation interleaved with optimizer steps);
    `load_state_dict` and a train -> eval transition always re-arm the check;
  * estimate of the whole batch's error = `sample_factor` x the sample's (a maximum over 32 x 512 values against one
    over 768 x 512 of the same distribution: 4.4 against 5.1 standard deviations);
  * estimate > `threshold` (0.7e-3: the contract with 30 % to spare) => every eval forward of this model runs the
    f32-class kernels from then on (2.2x the time, 1.4e-5) until a later check reads < 0.8 x threshold again;
  * every `recheck`-th forward repeats the comparison WITHOUT synchronising: copied rows, both small forwards and the
    reduction on the side stream, result through pinned memory, taken in by whichever later forward finds it done;
  * the near-tie refinement's probes (mining.RefinePolicy: fp16 against f32-class embeddings on 3 x slots rows of every
    `select_triplets` call) feed the same decision between checks.

`DeepSpeakerModel(..., f16_guard=None)` switches it off (raw fp16 whatever the network); `model.f16_guard.report()` and
bench.py's `precision_guard` object say what it saw and did.  Inside a stream capture nothing is measured (a capture
cannot synchronise): the verdict standing at capture time is captured; call `model.calibrate_precision(x)` first.
"""
from __future__ import annotations

from typing import Optional

import torch

F16_GUARD_THRESHOLD = 0.7e-3        # estimated max |d| / max |ref| above which precision "f16" forwards run f32-class
F16_GUARD_ROWS = 32
F16_GUARD_SAMPLE_FACTOR = 1.25
F16_GUARD_PROBE_FACTOR = 1.30       # the refinement's probes are fewer rows (3 x slots, typically 12)
F16_GUARD_RECHECK = 256             # forwards between asynchronous re-checks
F16_GUARD_MIN_GAP = 32              # forwards between synchronous checks triggered by a weight change
ESCALATED = "bf16x3"


class F16Guard:
    def __init__(self, threshold: float = F16_GUARD_THRESHOLD, rows: int = F16_GUARD_ROWS,
                 sample_factor: float = F16_GUARD_SAMPLE_FACTOR, recheck: int = F16_GUARD_RECHECK,
                 min_gap: int = F16_GUARD_MIN_GAP):
        self.threshold, self.rows, self.sample_factor = float(threshold), int(rows), float(sample_factor)
        self.recheck, self.min_gap = int(recheck), int(min_gap)
        self.verdict = "f16"
        self.key = None                 # weight generation of the last SYNCHRONOUS check
        self.calls = 0
        self.last_check = -(1 << 30)    # `calls` at the last check of either kind
        self.sample_error = None        # last measured max |e16 - e32| / max |e32| on the sample rows
        self.estimate = None            # sample_factor x that
        self.source = None              # what the standing verdict came from: "check", "recheck", "probes"
        self.checks = self.rechecks = self.escalations = self.deescalations = 0
        self.pending = None             # an asynchronous re-check in flight
        self._below = 0                 # consecutive measurements below the de-escalation bar while escalated
        self._seen = None               # weight generation the previous forward saw
        self._changed_last = False      # ... and whether THAT forward had seen a change too (weights changing every call)
        self._samples = 0               # checks of either kind so far: moves the sample's first row

    def invalidate(self):
        """New weights were loaded (load_state_dict) or the model came back from training: the next eval forward measures,
        whatever the rate limit says (a warm-up forward followed by a checkpoint load, or a sweep over
        checkpoints with fewer than `min_gap` forwards each, was never measured)."""
        self.last_check = -(1 << 30)
        self._changed_last = False

    def _sample(self, x: torch.Tensor, lengths=None):
        """`rows` utterances spread over the whole batch (not its head: a batch is anchors | positives | negatives, or
        sorted by length), the first of them moving with every check."""
        n = min(self.rows, x.shape[0])
        stride = max(1, x.shape[0] // n)
        off = self._samples % stride
        self._samples += 1
        xs = x[off::stride][:n]
        ls = None if lengths is None else lengths[off::stride][:n].contiguous()
        return xs, ls

    # ---- decisions --------------------------------------------------------------------------------------------
    def _decide(self, err: float, source: str, factor: Optional[float] = None):
        self.sample_error = float(err)
        self.estimate = float(err) * (self.sample_factor if factor is None else factor)
        if self.verdict == "f16" and self.estimate > self.threshold:
            self.verdict, self.source, self._below = ESCALATED, source, 0
            self.escalations += 1
        elif self.verdict == ESCALATED and source != "probes":
            # back to fp16: at once when a check on NEW weights reads below 0.8 x the threshold; an asynchronous re-check
            # (same weights: the verdict may have come from the refinement's probes, which see other rows) has to read
            # below it twice in a row -- no flapping between the two arithmetics every `recheck` forwards
            self._below = self._below + 1 if self.estimate < 0.8 * self.threshold else 0
            if self._below >= (1 if source == "check" else 2):
                self.verdict, self.source, self._below = "f16", source, 0
                self.deescalations += 1
        elif self.source is None:
            self.source = source

    def observe_probe(self, emb_err: Optional[float]):
        """mining.RefinePolicy.observe: a call's fp16-vs-f32-class embedding error on its sampled rows.  Only ever
        escalates (once escalated the probes compare the f32-class path with itself and read zero)."""
        if emb_err is not None and self.verdict == "f16":
            if float(emb_err) * F16_GUARD_PROBE_FACTOR > self.threshold:
                self._decide(float(emb_err), "probes", F16_GUARD_PROBE_FACTOR)

    # ---- measurements -----------------------------------------------------------------------------------------
    def _compare(self, model, eng, xs, lengths, stream_of):
        """enqueue (current stream): both forwards of xs and the reduction; returns the device pair [max |d|, max |ref|]"""
        folded = model._folded()
        e16 = eng.forward_eval_planned(xs, model._packed(with_f16=True), folded, precision="f16", lengths=lengths)
        e32 = eng.forward_eval_planned(xs, model._packed(with_bf16=True), folded, precision=ESCALATED, lengths=lengths)
        out = torch.empty(2, dtype=torch.float32, device=xs.device)
        eng.lib.call("ds_max_abs_diff_f32", eng._p(e16), eng._p(e32), e16.numel(), eng._p(out), eng._stream(stream_of))
        return out

    def calibrate(self, model, x: torch.Tensor, lengths=None) -> float:
        """Synchronous check on the first `rows` utterances of x (zero-padded batch if `lengths`); returns the sample's
        error.  One host synchronisation."""
        from .model import get_engine
        eng = get_engine()
        xs, ls = self._sample(x, lengths)
        xs = xs.contiguous().float()
        # the banks of both paths and the folded BatchNorm are built here, on the caller's stream
        out = self._compare(model, eng, xs, ls, xs)
        d, m = out.tolist()                                   # the synchronisation
        self.key = (model._pack_key, model._fold_key)
        self.last_check = self.calls
        self.checks += 1
        self._decide(self._ratio(d, m), "check")
        return self.sample_error

    @staticmethod
    def _ratio(d: float, m: float) -> float:
        # a non-finite pair (NaN / inf rows in either path) is an infinite error: escalate
        if d != d or m != m or d == float("inf") or m == float("inf"):
            return float("inf")
        return d / m if m > 0 else 0.0

    def _recheck_async(self, model, x: torch.Tensor):
        """The same comparison without a synchronisation: the sample rows are COPIED on the caller's stream (the caller
        may overwrite x), the work runs on the side stream, the pair lands in pinned memory."""
        from .mining import _side_stream
        from .model import get_engine
        eng = get_engine()
        xs = self._sample(x)[0].float().clone(memory_format=torch.contiguous_format)
        model._packed(with_bf16=True)           # (packed on the caller's stream if they do not exist yet)
        model._packed(with_f16=True)
        model._folded()
        main, side = torch.cuda.current_stream(x.device), _side_stream(x.device)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            xs.record_stream(side)
            out = self._compare(model, eng, xs, None, xs)
            host = torch.empty(2, dtype=torch.float32).pin_memory()
            host.copy_(out, non_blocking=True)
            ev = side.record_event()
        self.pending = {"event": ev, "host": host, "keep": (xs, out)}
        self.last_check = self.calls
        self.rechecks += 1

    def _collect(self):
        p = self.pending
        if p is not None and p["event"].query():
            d, m = p["host"].tolist()
            self.pending = None
            self._decide(self._ratio(d, m), "recheck")

    # ---- the question DeepSpeakerModel.forward asks ----------------------------------------------------------------
    def precision_for(self, model, x: torch.Tensor, lengths=None) -> str:
        self.calls += 1
        if x.is_cuda and torch.cuda.is_current_stream_capturing():
            return self.verdict                 # nothing can be measured inside a capture
        self._collect()
        model._packed(with_f16=True)            # refreshes the generation keys (cached objects otherwise)
        model._folded()
        key = (model._pack_key, model._fold_key)
        changed = self._seen is not None and key != self._seen
        # rate-limited only while the weights change on every forward (this one and the one before it both saw new
        # weights): a single change -- a checkpoint loaded after a warm-up forward -- is measured at once
        every_call = changed and self._changed_last
        self._seen, self._changed_last = key, changed
        if self.key is None or (key != self.key and (not every_call or self.calls - self.last_check >= self.min_gap)):
            self.calibrate(model, x, lengths)
        elif self.pending is None and self.calls - self.last_check >= self.recheck and lengths is None:
            self._recheck_async(model, x)
        return self.verdict

    def report(self) -> dict:
        return {"verdict": self.verdict, "threshold": self.threshold, "sample_error": self.sample_error,
                "estimated_error": self.estimate, "sample_rows": self.rows, "sample_factor": self.sample_factor,
                "source": self.source, "checks": self.checks, "rechecks": self.rechecks,
                "escalations": self.escalations, "deescalations": self.deescalations, "forwards": self.calls}
