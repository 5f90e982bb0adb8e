"""Triplet selection on the device.

`select_triplets` is the reference's per-triplet filter (train_triplet.py:249-274): keep triplet i
iff d(a_i,n_i) - d(a_i,p_i) < margin, indices ascending as `np.where` returns them.  The reference
round-trips embeddings AND raw inputs through NumPy on the host to do this; here the distances,
the mask and the ordered compaction are kernels and only the selected-row count crosses PCIe.
"""
from __future__ import annotations

import torch

from .model import _require_cuda, get_engine


# Near-tie refinement of the fp16 forward (precision "f16").  The fp16 path's embeddings are 3.7e-4 from the
# reference, which moves d_n - d_p by up to ~6e-4 (measured at the 768-utterance bench configuration; rms 2e-4):
# a triplet whose |d_n - d_p - margin| is below that can land on the other side of the filter.  Every triplet
# inside REFINE_BAND is therefore re-embedded through the split-operand bf16 path (f32-class, 5e-6) and decided
# on those distances; outside the band the fp16 decision is already the reference's.
REFINE_BAND = 1.25e-3       # 2x the largest observed |error| of d_n - d_p (6 sigma)
REFINE_CAP = 4              # near ties refined per call (expected ~1 per 256 random-init triplets; P(>4) = 0.3 %)


class TripletSelection:
    """Result of `select_triplets`.  Everything stays on the device; reading `.indices` / `.n_selected` /
    `.n_correct` is what synchronises (the reference branches on the count, train_triplet.py:263).

    When the fp16 forward's near ties were refined, the refinement ran on a side stream: every accessor first
    makes the CURRENT stream wait for it (no host synchronisation), so results are ordered like any other tensor."""

    def __init__(self, idx_full, count, d_p, d_n, mean_diff, loss=None, amb_count=None, amb_cap=0, ready=None):
        self._idx_full = idx_full    # int64 [N]; the first `count` entries are valid, ascending
        self._count = count          # int32 [1] on the device
        self._d_p = d_p              # [N]  (train_triplet.py:251)
        self._d_n = d_n              # [N]  (train_triplet.py:252)
        self._mean_diff = mean_diff  # mean(d_n - d_p), 1-element device tensor (train_triplet.py:259-260)
        self._loss = loss            # triplet loss on the same distances (train_triplet.py:275), 1-element tensor
        self._amb_count = amb_count  # near ties found (int32 [1]) when the fp16 forward was refined, else None
        self.amb_cap = amb_cap
        self._ready = ready          # event on the refinement stream, or None

    def wait(self):
        """Order the current stream after the refinement (idempotent per stream; free when nothing was refined)."""
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
    def refine_overflow(self) -> bool:
        """True if more near ties were found than the refinement had slots to re-embed (synchronises)."""
        return self._amb_count is not None and int(self.amb_count.item()) > self.amb_cap


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


def select_triplets(out_a: torch.Tensor, out_p: torch.Tensor, out_n: torch.Tensor, margin: float,
                    model=None, inputs=None, band: float = REFINE_BAND, cap: int = REFINE_CAP,
                    side_stream: bool = True) -> TripletSelection:
    """train_triplet.py:251-262.  With `model` (a DeepSpeakerModel in eval mode, precision "f16") and `inputs`
    (the three input batches the embeddings came from), near ties are re-embedded at f32-class precision first,
    which makes the selection the reference's (see REFINE_BAND).  No host synchronisation either way.  The
    re-embedding is a small-batch forward (latency-bound: a few workgroups walking the whole contraction), so by
    default it runs on a side stream next to whatever the caller enqueues next; `TripletSelection` orders its
    consumers after it."""
    _require_cuda(out_a, "select_triplets")
    eng = get_engine()
    a, p, n = (t.detach().contiguous() for t in (out_a, out_p, out_n))
    refine = model is not None and getattr(model, "precision", None) == "f16" and not model.training
    if not refine:
        t = eng.triplet_tail(a, p, n, margin)
        return TripletSelection(t["idx"], t["count"], t["d_p"], t["d_n"], t["mean_diff"], t["loss"])
    if inputs is None or len(inputs) != 3:
        raise ValueError("refinement needs inputs=(data_a, data_p, data_n), the batches behind the embeddings")
    t = eng.triplet_tail(a, p, n, margin, band=band, amb_cap=cap)
    main = torch.cuda.current_stream(a.device)
    side = _side_stream(a.device) if side_stream else main
    if side_stream:
        side.wait_stream(main)
    with torch.cuda.stream(side):
        if side_stream:
            for v in t.values():                    # the memoised main-stream buffers the side stream reads
                if isinstance(v, torch.Tensor):
                    v.record_stream(side)
        rows = inputs[0][0].numel()
        xr = torch.empty((3 * cap,) + tuple(inputs[0].shape[1:]), dtype=torch.float32, device=a.device)
        st = eng._stream(a)
        for k, x in enumerate(inputs):
            _require_cuda(x, "select_triplets(inputs=...)")
            x = x.contiguous()
            if side_stream:
                x.record_stream(side)               # keep the batch's memory until the side stream has read it
            eng.lib.call("ds_gather_rows_f32", eng._p(x), eng._p(t["amb_idx"]), eng._p(xr[k * cap:(k + 1) * cap]), cap,
                         rows, st)
        e_ref = model.embed_reference(xr)
        d_p, d_n = t["d_p"].clone(), t["d_n"].clone()       # the memoised fp16 distances stay what they are
        eng.lib.call("ds_refine_distances_f32", eng._p(e_ref), eng._p(t["amb_idx"]), eng._p(t["amb_count"]), cap,
                     eng._p(d_p), eng._p(d_n), a.shape[1], st)
        idx, count = torch.empty_like(t["idx"]), torch.empty_like(t["count"])
        mean_diff, loss = torch.empty_like(t["mean_diff"]), torch.empty_like(t["loss"])
        eng.lib.call("ds_triplet_scan_f32", eng._p(d_p), eng._p(d_n), float(margin), eng._p(loss), eng._p(idx),
                     eng._p(count), eng._p(mean_diff), d_p.numel(), st)
        ready = side.record_event() if side_stream else None
    return TripletSelection(idx, count, d_p, d_n, mean_diff, loss, t["amb_count"], cap, ready)


class MinedNegatives:
    """Result of a search that was enqueued on the side stream: `indices` / `distances` order the calling stream
    after the search before they hand out the tensors; unpacks like the plain (indices, distances) tuple."""

    def __init__(self, idx, dist, ready):
        self._idx, self._dist, self._ready = idx, dist, ready

    def wait(self):
        if self._ready is not None:
            torch.cuda.current_stream(self._idx.device).wait_event(self._ready)
            self._ready = None
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
