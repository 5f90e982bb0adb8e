"""Triplet selection on the device.

`select_triplets` is the reference's per-triplet filter (train_triplet.py:249-274): keep triplet i
iff d(a_i,n_i) - d(a_i,p_i) < margin, indices ascending as `np.where` returns them.  The reference
round-trips embeddings AND raw inputs through NumPy on the host to do this; here the distances,
the mask and the ordered compaction are kernels and only the selected-row count crosses PCIe.
"""
from __future__ import annotations


import torch

from .model import _require_cuda, get_engine


# Near-tie refinement of the fp16 forward (precision "f16").  The fp16 path's embeddings are a few 1e-4 from the
# reference, which moves the filter's decision variable d_n - d_p by ~6e-4 at the random-init bench configuration
# (rms 2e-4) -- and by more once the weights spread the embeddings apart: the error of a distance scales with the
# distance.  A triplet whose |d_n - d_p - margin| is below that error can land on the other side of the filter.  Every
# triplet inside a BAND around the boundary is therefore re-embedded through the split-operand bf16 path (f32-class,
# 5e-6) and decided on those distances; outside the band the fp16 decision is already the reference's -- provided the
# band really covers the error.  Round 4: the band is no longer a constant calibrated on one input distribution, it is
# MEASURED, continuously and for free:
#
#   * the refinement forward re-embeds all its slots whether near ties fill them or not, so the scan puts PROBE
#     triplets (a window that rotates through the batch from call to call) into the slots near ties leave unused;
#   * the kernel that patches the distances reports max |(d_n - d_p)_f32-class - (d_n - d_p)_fp16| over all slots --
#     the fp16 error on near ties and probes alike -- and the value travels back through pinned memory with the
#     near-tie count (no synchronisation in the step);
#   * `RefinePolicy.band_for()` = max(REFINE_BAND, BAND_SAFETY x the largest error seen in the last BAND_WINDOW calls);
#   * a call whose own samples show an error above HALF the band it used is a band violation: its `TripletSelection`
#     re-embeds the WHOLE batch at f32-class precision before it hands out anything, exactly like a slot overflow
#     (`band_exceeded`, `refined_all`), and the next calls use the wider band.
#
# How many near ties there are depends on the weights: ~1 per 256 at random init, but a triplet-trained network
# concentrates d_n - d_p AT the margin.  And a slot is not free: a re-embedded triplet is 3 utterances x 3 MFMAs per
# product = 9 fp16 utterance-forwards of matrix work (measured on the 768-utterance step: 4 slots +0 %, 8 +4 %,
# 16 +10 %, 32 +24 %).  The number of re-embedding slots is therefore neither a constant nor generous by default:
#   * `RefinePolicy` sizes the slots of call k from the near-tie counts of the calls before it (read back through
#     pinned memory, polled without blocking): REFINE_CAP_START until it has seen a few calls, then twice the recent
#     maximum in powers of two, from REFINE_CAP_MIN up to the whole batch (at which point the "refinement" simply is
#     the f32-class forward of the batch);
#   * if a call still finds more near ties than it had slots for, its `TripletSelection` re-embeds the WHOLE batch at
#     f32-class precision before it hands out anything (`refined_all`), and the policy has learned the new level.
# So what an accessor returns is always decided on f32-class distances inside the band, whatever the weights are.
REFINE_BAND = 1.25e-3       # the band's floor and starting value (2x the largest |error| of d_n - d_p at random init)
BAND_SAFETY = 2.5           # band >= this x the largest error observed in the window
BAND_VIOLATION = 0.5        # a call observing an error above this x its band falls back to the whole batch
BAND_WINDOW = 512           # calls the observed-error maximum is taken over
PROBE_STRIDE = 37           # the probe window moves by this many triplets per call
REFINE_CAP_START = 32       # slots while the policy has no history (96 re-embedded rows)
REFINE_CAP_MIN = 4          # smallest slot count once it has
REFINE_CAP = REFINE_CAP_START


def _pow2ceil(v: int) -> int:
    return 1 << max(0, int(v) - 1).bit_length()


class RefinePolicy:
    """Host-side sizing of the near-tie refinement (one per model): slot count and band for the next call from the
    near-tie counts and fp16 errors observed so far.  Pure bookkeeping -- no device work, no synchronisation."""

    HISTORY = 8
    RING = 256                          # pinned read-back slots (one per call in flight)

    WARM = 4                            # observed calls before the slot count may drop below cap_start

    def __init__(self, cap_min: int = REFINE_CAP_MIN, cap_start: int = REFINE_CAP_START,
                 band_floor: float = REFINE_BAND, band_safety: float = BAND_SAFETY):
        self.cap_min, self.cap_start = cap_min, max(cap_min, cap_start)
        self.band_floor, self.band_safety = float(band_floor), float(band_safety)
        self.n_seen = 0
        self.seen = []                  # near-tie counts of the last HISTORY observed calls
        self.errs = []                  # per-call max fp16 error of d_n - d_p, last BAND_WINDOW calls that sampled any
        self.err_samples = 0            # slots sampled so far (near ties + probes)
        self.err_max_ever = 0.0
        self.emb_errs = []              # per-call embedding error of the fp16 path on the sampled rows (max |d| / max |ref|)
        self.band_violations = 0
        self.pending = []               # read-back records of calls whose count has not been taken in yet
        self.window = 1                 # calls whose near ties share ONE refinement forward (see RefineWindow)
        self.batch = None               # the RefineWindow collecting calls, if window > 1
        self.calls = self.overflows = 0
        self.max_seen = 0
        self._ring = self._ring_err = None
        self._next = 0
        self.guard = None               # the model's precision_guard.F16Guard (refine_policy() sets it)

    def observe(self, count: int, err: float = None, n_samples: int = 0, emb_err: float = None):
        self.seen = (self.seen + [int(count)])[-self.HISTORY:]
        self.max_seen = max(self.max_seen, int(count))
        self.n_seen += 1
        if err is not None and n_samples > 0:
            self.errs = (self.errs + [float(err)])[-BAND_WINDOW:]
            self.err_samples += int(n_samples)
            self.err_max_ever = max(self.err_max_ever, float(err))
        if emb_err is not None:
            self.emb_errs = (self.emb_errs + [float(emb_err)])[-BAND_WINDOW:]
            if self.guard is not None:          # the model's fp16 precision guard takes the same sample in
                self.guard.observe_probe(emb_err)

    @property
    def embedding_error_observed(self) -> float:
        """Largest embedding error (max |fp16 - f32-class| / max |f32-class| over a call's 3 x slots sampled rows) of the
        last BAND_WINDOW calls: the fp16 path's measured distance to north_star's 1e-3 on whatever it has been embedding
        (0.0 before any call sampled).  On seeded weights 4e-4 - 5e-4; a trained network with spread embeddings reaches
        1e-3 -- `precision="bf16x3"` is the answer when this says so."""
        return max(self.emb_errs, default=0.0)

    @property
    def err_max_window(self) -> float:
        return max(self.errs, default=0.0)

    def band_for(self) -> float:
        """The band of the next call: never below the floor, and BAND_SAFETY x the largest fp16 error of d_n - d_p
        that any slot (near tie or probe) of the last BAND_WINDOW calls has shown."""
        return max(self.band_floor, self.band_safety * self.err_max_window)

    def readback(self, amb_count: torch.Tensor, err: torch.Tensor = None) -> dict:
        """Enqueue (on the current stream) the copy of a call's near-tie count (and its observed error pair) into a
        pinned slot; the returned record carries the event after which `value()` is valid."""
        if self._ring is None:
            self._ring = torch.zeros(self.RING, dtype=torch.int32).pin_memory()
            self._ring_err = torch.zeros((self.RING, 5), dtype=torch.float32).pin_memory()
        slot = self._next % self.RING
        self._next += 1
        if err is not None and err.numel() == 5:
            # ds_refine_distances_fused_f32 put the near-tie count into err[4]: ONE copy brings everything back (beside the
            # persistent convolutions every side-stream operation waits for the drain of a main-stream launch)
            self._ring_err[slot].copy_(err, non_blocking=True)
            fused = True
        else:
            fused = False
            self._ring[slot:slot + 1].copy_(amb_count, non_blocking=True)
            if err is not None:
                self._ring_err[slot, :4].copy_(err, non_blocking=True)
        rec = {"slot": slot, "serial": self._next, "event": torch.cuda.current_stream(amb_count.device).record_event(),
               "device": amb_count, "device_err": err, "taken": False, "fused": fused}
        self.pending.append(rec)
        return rec

    def value(self, rec: dict):
        """(near-tie count, observed error or None, slots sampled) of a finished call (its event must have completed)."""
        if self._next - rec["serial"] >= self.RING:     # the slot has been handed to a later call since
            cnt = int(rec["device"].item())
            e = rec["device_err"].tolist() if rec["device_err"] is not None else None
        else:
            e = self._ring_err[rec["slot"]].tolist() if rec["device_err"] is not None else None
            # fused read-back (ds_refine_distances_fused_f32): the count travelled as err[4]
            cnt = int(round(e[4])) if rec.get("fused") else int(self._ring[rec["slot"]])
        if e is None:
            return (cnt, None, 0, None)
        return (cnt, float(e[0]), int(e[1]), (float(e[2]) / float(e[3])) if e[3] > 0 else None)

    def take(self, rec: dict):
        v = self.value(rec)
        if not rec["taken"]:
            rec["taken"] = True
            self.observe(*v)
        return v

    def poll(self):
        """Take in the counts of earlier calls whose device work has finished (never blocks)."""
        still = []
        for rec in self.pending:
            if rec["taken"]:
                continue
            if rec["event"].query():
                self.take(rec)
            else:
                still.append(rec)
        self.pending = still[-(self.RING // 2):]

    def flush(self):
        """Run the refinement forward of the calls collected so far (no-op when nothing is pending): what a caller does
        at the end of a timed region, or before it changes the weights."""
        if self.batch is not None:
            self.batch.flush()

    def cap_for(self, n_triplets: int) -> int:
        """Slots for a batch of n_triplets: twice the recent maximum, a power of two, at least cap_min (cap_start
        while fewer than WARM calls have been observed); once that passes half the batch the whole batch is
        re-embedded (cap == n_triplets)."""
        floor = self.cap_min if self.n_seen >= self.WARM else self.cap_start
        want = max(floor, _pow2ceil(2 * max(self.seen, default=0)))
        return n_triplets if want > n_triplets // 2 else want


class RefineWindow:
    """Near ties of up to `window` consecutive `select_triplets` calls re-embedded by ONE f32-class forward.

    Why: the refinement forward is ~16 launches of a few workgroups each; next to persistent kernels that hold every
    SIMD's registers they do not overlap with the following forward, they interleave with it -- each side-stream launch
    delays one main-stream launch by its own length (measured: +0.21 ms on a 1.80 ms forward at 4 slots, whatever the
    slot count below ~16).  The rows of K calls through the same launches cost the same once instead of K times.

    A call adds its near-tie utterances (gathered on ITS stream into the window's buffer: one launch) and returns a
    `TripletSelection` that is still open; the K-th call -- or whoever reads an open selection first, or
    `RefinePolicy.flush()` -- runs the forward on the side stream and completes all of them.  Results are exactly those
    of per-call refinement: the same rows through the same kernels (per-row results of the eval forward do not depend on
    the batch they ride in)."""

    def __init__(self, policy, eng, cap, row_shape, device, pw_ref, folded_ref, window, side_stream):
        self.policy, self.eng, self.cap, self.row_shape = policy, eng, cap, tuple(row_shape)
        self.pw_ref, self.folded_ref, self.window, self.side = pw_ref, folded_ref, window, side_stream
        self.xr = torch.empty((window * 3 * cap,) + self.row_shape, dtype=torch.float32, device=device)
        self.entries = []

    def matches(self, cap, row_shape, pw_ref, folded_ref, side_stream) -> bool:
        return (cap == self.cap and tuple(row_shape) == self.row_shape and pw_ref is self.pw_ref
                and folded_ref is self.folded_ref and side_stream == self.side)

    def add(self, t, xs, a, p, n, margin, sel):
        eng, cap, k = self.eng, self.cap, len(self.entries)
        rows = xs[0][0].numel()
        eng.lib.call("ds_gather_rows3_f32", eng._p(xs[0]), eng._p(xs[1]), eng._p(xs[2]), eng._p(t["amb_idx"]),
                     eng._p(self.xr[k * 3 * cap:(k + 1) * 3 * cap]), cap, rows, eng._stream(a))
        main = torch.cuda.current_stream(a.device)
        self.entries.append({"t": t, "a": a, "p": p, "n": n, "margin": float(margin), "sel": sel,
                             "gathered": main.record_event(), "stream": main})
        if len(self.entries) >= self.window:
            self.flush()

    def flush(self):
        ents, self.entries = self.entries, []
        if not ents:
            return
        if self.policy.batch is self:
            self.policy.batch = None            # this window's buffer is in flight: the next call opens a new one
        eng, cap = self.eng, self.cap
        dev = self.xr.device
        cur = torch.cuda.current_stream(dev)
        side = _side_stream(dev) if self.side else cur
        for en in ents:
            side.wait_event(en["gathered"])
        with torch.cuda.stream(side):
            if side != cur or any(en["stream"] != side for en in ents):
                self.xr.record_stream(side)
                for en in ents:
                    for v in list(en["t"].values()) + [en["a"], en["p"], en["n"]]:
                        if isinstance(v, torch.Tensor):
                            v.record_stream(side)
            st = eng._stream(self.xr)
            e_ref = eng.forward_eval_planned(self.xr[:len(ents) * 3 * cap], self.pw_ref, self.folded_ref, precision="bf16x3")
            for k, en in enumerate(ents):
                t, a, p, n = en["t"], en["a"], en["p"], en["n"]
                e_k = e_ref[k * 3 * cap:(k + 1) * 3 * cap]
                d_p, d_n = torch.empty_like(t["d_p"]), torch.empty_like(t["d_n"])
                err = torch.empty(5, dtype=torch.float32, device=dev)
                eng.lib.call("ds_refine_distances_fused_f32", eng._p(e_k), eng._p(t["amb_idx"]), eng._p(t["amb_count"]), cap,
                             eng._p(d_p), eng._p(d_n), eng._p(t["d_p"]), eng._p(t["d_n"]), eng._p(a), eng._p(p), eng._p(n),
                             d_p.numel(), a.shape[1], eng._p(err), st)
                rb = self.policy.readback(t["amb_count"], err)
                idx, count = torch.empty_like(t["idx"]), torch.empty_like(t["count"])
                mean_diff, loss = torch.empty_like(t["mean_diff"]), torch.empty_like(t["loss"])
                eng.lib.call("ds_triplet_scan_f32", eng._p(d_p), eng._p(d_n), en["margin"], eng._p(loss), eng._p(idx),
                             eng._p(count), eng._p(mean_diff), d_p.numel(), st)
                en["sel"]._complete(idx, count, d_p, d_n, mean_diff, loss, rb)
            ready = side.record_event()
        for en in ents:             # (also when the forward ran on the flushing caller's own stream: a selection of the
            en["sel"]._ready = ready    # window may be read from another stream than the one that closed it)


class TripletSelection:
    """Result of `select_triplets`.  Everything stays on the device; reading `.indices` / `.n_selected` /
    `.n_correct` is what synchronises (the reference branches on the count, train_triplet.py:263).

    When the fp16 forward's near ties were refined, the refinement ran on a side stream: every accessor first makes
    the CURRENT stream wait for it.  Accessors of a refined selection also check (pinned values, after the side
    stream's event) that the refinement had a slot for every near tie AND that the fp16 error its slots observed stays
    below half the band it used; if not, the whole batch is re-embedded at f32-class precision first (`refined_all`),
    so what they return never rests on an undecided fp16 comparison."""

    def __init__(self, idx_full, count, d_p, d_n, mean_diff, loss=None, amb_count=None, amb_cap=0, ready=None,
                 readback=None, fallback=None, policy=None, band=0.0):
        self._idx_full = idx_full    # int64 [N]; the first `count` entries are valid, ascending
        self._count = count          # int32 [1] on the device
        self._d_p = d_p              # [N]  (train_triplet.py:251)
        self._d_n = d_n              # [N]  (train_triplet.py:252)
        self._mean_diff = mean_diff  # mean(d_n - d_p), 1-element device tensor (train_triplet.py:259-260)
        self._loss = loss            # triplet loss on the same distances (train_triplet.py:275), 1-element tensor
        self._amb_count = amb_count  # near ties found (int32 [1]) when the fp16 forward was refined, else None
        self.amb_cap = amb_cap
        self.band = band             # the band this call used
        self._ready = ready          # event on the refinement stream, or None
        self._readback = readback    # RefinePolicy.readback record of amb_count (pinned slot + its event)
        self._n_amb = None
        self.embedding_error = None
        self._err = None             # (max observed fp16 error of d_n - d_p over this call's slots, slots sampled)
        self._fallback = fallback    # () -> dict of replacement tensors: the whole batch at f32-class precision
        self._policy = policy
        self._window = None          # the RefineWindow this selection is still waiting in (select_triplets(window > 1))
        self._resolved = fallback is None or readback is None
        self.refined_all = False     # True once the overflow / band-violation action has replaced the results
        self.band_exceeded = False   # True if this call's own samples showed an error above BAND_VIOLATION x its band

    def _complete(self, idx, count, d_p, d_n, mean_diff, loss, readback):
        """RefineWindow.flush: the refined results of a selection that was opened with the fp16 ones."""
        self._idx_full, self._count, self._d_p, self._d_n, self._mean_diff, self._loss = idx, count, d_p, d_n, mean_diff, loss
        self._readback = readback
        self._window = None

    def _close(self):
        if self._window is not None:
            self._window.flush()        # completes every selection of the window, this one included

    def _take(self):
        self._close()
        if self._n_amb is None:
            self._readback["event"].synchronize()
            self._n_amb, err, n_s, emb_err = self._policy.take(self._readback)
            self._err = (err, n_s)
            self.embedding_error = emb_err      # fp16 vs f32-class embeddings on this call's sampled rows (None: not sampled)
        return self._n_amb

    def resolve(self):
        """Make sure every near tie was decided at f32-class precision and that the band was wide enough (see the
        class docstring).  Waits on the host for the refinement's event -- the same wait any host-side read of the
        selection implies."""
        self._close()
        if not self._resolved:
            self._resolved = True
            n_amb = self._take()
            err = self._err[0]
            self.band_exceeded = err is not None and err > BAND_VIOLATION * self.band
            if n_amb > self.amb_cap or self.band_exceeded:
                if n_amb > self.amb_cap:
                    self._policy.overflows += 1
                if self.band_exceeded:
                    self._policy.band_violations += 1
                t = self._fallback()
                self._idx_full, self._count, self._d_p, self._d_n = t["idx"], t["count"], t["d_p"], t["d_n"]
                self._mean_diff, self._loss = t["mean_diff"], t["loss"]
                self._ready = None                  # produced on the current stream
                self.refined_all = True
            self._fallback = None                   # drops the references to the input batches
        return self

    def wait(self):
        """Order the current stream after the refinement (idempotent; free when nothing was refined)."""
        self.resolve()
        if self._ready is not None:
            cur = torch.cuda.current_stream(self._idx_full.device)
            cur.wait_event(self._ready)
            for t in (self._idx_full, self._count, self._d_p, self._d_n, self._mean_diff, self._loss):
                if t is not None:
                    t.record_stream(cur)        # allocated on the side stream, consumed on this one
        return self

    count = property(lambda self: self.wait()._count)
    d_p = property(lambda self: self.wait()._d_p)
    d_n = property(lambda self: self.wait()._d_n)
    mean_diff = property(lambda self: self.wait()._mean_diff)
    loss = property(lambda self: self.wait()._loss)
    amb_count = property(lambda self: self.wait()._amb_count)

    @property
    def n_selected(self) -> int:
        return int(self.count.item())

    @property
    def indices(self) -> torch.Tensor:
        """int64 [n_selected], ascending -- `np.where(all == 1)[0]` of train_triplet.py:262."""
        n = self.n_selected
        return self._idx_full[:n]

    @property
    def n_correct(self) -> int:
        """triplets already satisfying the margin (train_triplet.py:256-257)"""
        return self._idx_full.numel() - self.n_selected

    @property
    def n_near_ties(self) -> int:
        """near ties the fp16 forward left undecided (0 when nothing was refined); synchronises"""
        if self._amb_count is None:
            return 0
        return self._take()

    @property
    def observed_error(self):
        """(largest |fp16 error of d_n - d_p| over this call's slots, slots sampled) or (None, 0); synchronises"""
        self._close()
        if self._amb_count is None or self._readback is None:
            return (None, 0)
        self._take()
        return self._err

    @property
    def refine_overflow(self) -> bool:
        """True if more near ties were found than the refinement had slots for -- in which case the results were
        replaced by the whole-batch f32-class ones (`refined_all`); synchronises."""
        return self.n_near_ties > self.amb_cap


_side_streams = {}


def side_stream(device) -> "torch.cuda.Stream":
    """The stream `select_triplets` refines near ties on and `mine_semihard_negatives(side_stream=True)` searches on
    (one per device): callers that feed these with asynchronous collectives let THIS stream wait for them."""
    return _side_stream(device)


def _side_stream(device) -> "torch.cuda.Stream":
    s = _side_streams.get(device)
    if s is None:
        s = _side_streams[device] = torch.cuda.Stream(device=device)
    return s


def refine_policy(model) -> RefinePolicy:
    pol = getattr(model, "_refine_policy", None)
    if pol is None:
        pol = RefinePolicy()
        pol.guard = getattr(model, "f16_guard", None)
        object.__setattr__(model, "_refine_policy", pol)
    return pol


def select_triplets(out_a: torch.Tensor, out_p: torch.Tensor, out_n: torch.Tensor, margin: float,
                    model=None, inputs=None, band: float = None, cap: int = None,
                    side_stream: bool = True, window: int = None) -> TripletSelection:
    """train_triplet.py:251-262.  With `model` (a DeepSpeakerModel in eval mode, precision "f16") and `inputs`
    (the three input batches the embeddings came from), near ties are re-embedded at f32-class precision first,
    which makes the selection the reference's (see REFINE_BAND).  The call itself never synchronises with the
    host.  The re-embedding is a small-batch forward (latency-bound: a few workgroups walking the whole
    contraction), so by default it runs on a side stream next to whatever the caller enqueues next;
    `TripletSelection` orders its consumers after it.  `cap`: re-embedding slots; default: sized by the model's
    `RefinePolicy` from the near-tie counts of earlier calls.  `band`: half-width of the near-tie band; default: the
    policy's, i.e. measured (the slots near ties leave unused carry probe triplets whose fp16 error is read back; a
    call that observes an error above half its band re-embeds the whole batch when the selection is read).
    `window` (default: the policy's, 1): the near ties of this many consecutive calls share one refinement forward
    (`RefineWindow`); a selection whose window is still open closes it when it is read."""
    _require_cuda(out_a, "select_triplets")
    eng = get_engine()
    a, p, n = (t.detach().contiguous() for t in (out_a, out_p, out_n))
    refine = model is not None and getattr(model, "precision", None) == "f16" and not model.training
    if not refine:
        t = eng.triplet_tail(a, p, n, margin)
        return TripletSelection(t["idx"], t["count"], t["d_p"], t["d_n"], t["mean_diff"], t["loss"])
    if inputs is None or len(inputs) != 3:
        raise ValueError("refinement needs inputs=(data_a, data_p, data_n), the batches behind the embeddings")
    xs = []
    for x in inputs:
        _require_cuda(x, "select_triplets(inputs=...)")
        if x.shape[0] != a.shape[0]:
            raise ValueError("inputs must hold one utterance per embedding row")
        xs.append(x.contiguous().float())           # ds_gather_rows_f32 copies float32 rows
    n_trip = a.shape[0]
    policy = refine_policy(model)
    policy.poll()
    policy.calls += 1
    cap = policy.cap_for(n_trip) if cap is None else max(1, min(int(cap), n_trip))
    band = policy.band_for() if band is None else float(band)
    whole = cap >= n_trip
    t = eng.triplet_tail(a, p, n, margin, band=band, amb_cap=cap,
                         probe_base=-1 if whole else (policy.calls * PROBE_STRIDE) % n_trip)
    main = torch.cuda.current_stream(a.device)
    side = _side_stream(a.device) if side_stream else main
    # the f32-class path's packed filters / folded BatchNorm are built HERE, on the caller's stream, if they do not
    # exist yet: built inside the side-stream section, a later main-stream use would not be ordered after the packing
    pw_ref = model._packed(with_bf16=True)
    folded_ref = model._folded()

    def embed_all():
        """the overflow action / the whole-batch tier: every triplet decided on f32-class embeddings.  Runs with the
        packed filters and folded BatchNorm of THIS call (pinned above): an overflow is resolved when the selection
        is read, possibly after an optimizer step or a train-mode forward has changed the model -- the replacement
        must come from the weights that produced the embeddings it replaces."""
        e = eng.forward_eval_planned(torch.cat(xs), pw_ref, folded_ref, precision="bf16x3")
        return eng.triplet_tail(*(r.contiguous() for r in e.split(n_trip)), margin)

    window = policy.window if window is None else int(window)
    if window > 1 and not whole:
        w = policy.batch
        if w is not None and not w.matches(cap, xs[0].shape[1:], pw_ref, folded_ref, side_stream):
            w.flush()                   # other slots / shapes / weights: close what is open, start afresh
            w = None
        if w is None:
            w = policy.batch = RefineWindow(policy, eng, cap, xs[0].shape[1:], a.device, pw_ref, folded_ref, window, side_stream)
        # opened with the fp16 results (never handed out: every accessor closes the window first)
        sel = TripletSelection(t["idx"], t["count"], t["d_p"], t["d_n"], t["mean_diff"], t["loss"], t["amb_count"], cap, None,
                               readback=None, fallback=embed_all, policy=policy, band=band)
        sel._window, sel._resolved = w, False
        w.add(t, xs, a, p, n, margin, sel)
        return sel
    if policy.batch is not None:
        policy.batch.flush()            # calls complete in order
    if side_stream:
        side.wait_stream(main)
    with torch.cuda.stream(side):
        if side_stream:
            for v in list(t.values()) + xs + [a, p, n]:     # main-stream memory the side stream reads
                if isinstance(v, torch.Tensor):
                    v.record_stream(side)
        if whole:
            rb = policy.readback(t["amb_count"])
            r = embed_all()
            idx, count, d_p, d_n, mean_diff, loss = r["idx"], r["count"], r["d_p"], r["d_n"], r["mean_diff"], r["loss"]
        else:
            rows = xs[0][0].numel()
            xr = torch.empty((3 * cap,) + tuple(xs[0].shape[1:]), dtype=torch.float32, device=a.device)
            st = eng._stream(a)
            # (round 6: one gather, one patch kernel that also copies the distances and carries the count, one read-back
            # copy -- 19 side-stream operations per call instead of 24)
            eng.lib.call("ds_gather_rows3_f32", eng._p(xs[0]), eng._p(xs[1]), eng._p(xs[2]), eng._p(t["amb_idx"]), eng._p(xr), cap,
                         rows, st)
            e_ref = eng.forward_eval_planned(xr, pw_ref, folded_ref, precision="bf16x3")
            d_p, d_n = torch.empty_like(t["d_p"]), torch.empty_like(t["d_n"])
            err = torch.empty(5, dtype=torch.float32, device=a.device)
            eng.lib.call("ds_refine_distances_fused_f32", eng._p(e_ref), eng._p(t["amb_idx"]), eng._p(t["amb_count"]), cap,
                         eng._p(d_p), eng._p(d_n), eng._p(t["d_p"]), eng._p(t["d_n"]), eng._p(a), eng._p(p), eng._p(n),
                         d_p.numel(), a.shape[1], eng._p(err), st)
            rb = policy.readback(t["amb_count"], err)
            idx, count = torch.empty_like(t["idx"]), torch.empty_like(t["count"])
            mean_diff, loss = torch.empty_like(t["mean_diff"]), torch.empty_like(t["loss"])
            eng.lib.call("ds_triplet_scan_f32", eng._p(d_p), eng._p(d_n), float(margin), eng._p(loss), eng._p(idx),
                         eng._p(count), eng._p(mean_diff), d_p.numel(), st)
        ready = side.record_event()
    return TripletSelection(idx, count, d_p, d_n, mean_diff, loss, t["amb_count"], cap, ready if side_stream else None,
                            readback=rb, fallback=None if whole else embed_all, policy=policy, band=band)


class MinedNegatives:
    """Result of a search that was enqueued on the side stream: `indices` / `distances` order the calling stream
    after the search before they hand out the tensors; unpacks like the plain (indices, distances) tuple."""

    def __init__(self, idx, dist, ready):
        self._idx, self._dist, self._ready = idx, dist, ready

    def wait(self):
        if self._ready is not None:
            cur = torch.cuda.current_stream(self._idx.device)
            cur.wait_event(self._ready)             # idempotent: the event stays for consumers on other streams
            self._idx.record_stream(cur)            # allocated on the side stream, consumed on this one
            self._dist.record_stream(cur)
        return self

    @property
    def indices(self) -> torch.Tensor:
        return self.wait()._idx

    @property
    def distances(self) -> torch.Tensor:
        return self.wait()._dist

    def __iter__(self):
        self.wait()
        return iter((self._idx, self._dist))


def mine_semihard_negatives(anchors: torch.Tensor, positives: torch.Tensor, anchor_labels: torch.Tensor,
                            candidates: torch.Tensor, candidate_labels: torch.Tensor, side_stream: bool = False):
    """Cross-GPU hard-negative search (BASELINE.json north_star; no reference counterpart).

    `candidates` / `candidate_labels` are normally the RCCL all-gather of every rank's embeddings and
    speaker ids.  For anchor i returns the index j of the candidate with a different speaker that is
    closest to the anchor among those farther than its positive (semi-hard), else the closest one;
    ties -> lowest j; -1 if no candidate has another speaker.  Also returns the distances d(a_i, x_j).

    `side_stream`: the mined negatives feed the NEXT batch, nothing of this step waits for them -- enqueue the search
    on the side stream (after everything enqueued so far) and return a `MinedNegatives`, so the caller's stream goes
    straight on to its next forward."""
    _require_cuda(anchors, "mine_semihard_negatives")
    eng = get_engine()
    a, p, c = (t.detach().contiguous() for t in (anchors, positives, candidates))
    la, lc = anchor_labels.to(torch.int64).contiguous(), candidate_labels.to(torch.int64).contiguous()
    main = torch.cuda.current_stream(a.device)
    side = _side_stream(a.device) if side_stream else main
    if side_stream:
        side.wait_stream(main)
        for t in (a, p, c, la, lc):
            t.record_stream(side)                   # main-stream memory the side stream reads
    with torch.cuda.stream(side):
        d_p = eng.pairwise_distance(a, p)
        n, d = a.shape
        idx = torch.empty(n, dtype=torch.int64, device=a.device)
        dist_out = torch.empty(n, dtype=torch.float32, device=a.device)
        ws = torch.empty(eng.lib.raw("ds_mine_workspace_floats")(n, c.shape[0]), dtype=torch.float32, device=a.device)
        eng.lib.call("ds_mine_semihard_f32", eng._p(a), eng._p(d_p), eng._p(la), eng._p(c), eng._p(lc), eng._p(ws),
                     eng._p(idx), eng._p(dist_out), n, c.shape[0], d, eng._stream(a))
        if side_stream:
            return MinedNegatives(idx, dist_out, side.record_event())
    return idx, dist_out
