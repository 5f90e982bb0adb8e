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
REFINE_BAND = 2e-3          # > 3x the largest observed |error| of d_n - d_p
REFINE_CAP = 8              # near ties refined per call (expected ~1.5 per 256 random-init triplets)


class TripletSelection:
    """Result of `select_triplets`.  Everything stays on the device; reading `.indices` / `.n_selected` /
    `.n_correct` is what synchronises (the reference branches on the count, train_triplet.py:263)."""

    def __init__(self, idx_full, count, d_p, d_n, mean_diff, loss=None, amb_count=None, amb_cap=0):
        self._idx_full = idx_full    # int64 [N]; the first `count` entries are valid, ascending
        self.count = count           # int32 [1] on the device
        self.d_p = d_p               # [N]  (train_triplet.py:251)
        self.d_n = d_n               # [N]  (train_triplet.py:252)
        self.mean_diff = mean_diff   # mean(d_n - d_p), 1-element device tensor (train_triplet.py:259-260)
        self.loss = loss             # triplet loss on the same distances (train_triplet.py:275), 1-element tensor
        self.amb_count = amb_count   # near ties found (int32 [1]) when the fp16 forward was refined, else None
        self.amb_cap = amb_cap

    @property
    def n_selected(self) -> int:
        return int(self.count.item())

    @property
    def indices(self) -> torch.Tensor:
        """int64 [n_selected], ascending -- `np.where(all == 1)[0]` of train_triplet.py:262."""
        return self._idx_full[:self.n_selected]

    @property
    def n_correct(self) -> int:
        """triplets already satisfying the margin (train_triplet.py:256-257)"""
        return self._idx_full.numel() - self.n_selected

    @property
    def refine_overflow(self) -> bool:
        """True if more near ties were found than `REFINE_CAP` slots could re-embed (synchronises)."""
        return self.amb_count is not None and int(self.amb_count.item()) > self.amb_cap


def select_triplets(out_a: torch.Tensor, out_p: torch.Tensor, out_n: torch.Tensor, margin: float,
                    model=None, inputs=None, band: float = REFINE_BAND, cap: int = REFINE_CAP) -> TripletSelection:
    """train_triplet.py:251-262.  With `model` (a DeepSpeakerModel in eval mode, precision "f16") and `inputs`
    (the three input batches the embeddings came from), near ties are re-embedded at f32-class precision first,
    which makes the selection the reference's (see REFINE_BAND); no host synchronisation either way."""
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
    rows = inputs[0][0].numel()
    xr = torch.empty((3 * cap,) + tuple(inputs[0].shape[1:]), dtype=torch.float32, device=a.device)
    st = eng._stream(a)
    for k, x in enumerate(inputs):
        _require_cuda(x, "select_triplets(inputs=...)")
        x = x.contiguous()
        eng.lib.call("ds_gather_rows_f32", eng._p(x), eng._p(t["amb_idx"]), eng._p(xr[k * cap:(k + 1) * cap]), cap, rows, st)
    e_ref = model.embed_reference(xr)
    d_p, d_n = t["d_p"].clone(), t["d_n"].clone()       # the memoised fp16 distances stay what they are
    eng.lib.call("ds_refine_distances_f32", eng._p(e_ref), eng._p(t["amb_idx"]), eng._p(t["amb_count"]), cap,
                 eng._p(d_p), eng._p(d_n), a.shape[1], st)
    idx, count = torch.empty_like(t["idx"]), torch.empty_like(t["count"])
    mean_diff, loss = torch.empty_like(t["mean_diff"]), torch.empty_like(t["loss"])
    eng.lib.call("ds_triplet_scan_f32", eng._p(d_p), eng._p(d_n), float(margin), eng._p(loss), eng._p(idx), eng._p(count),
                 eng._p(mean_diff), d_p.numel(), st)
    return TripletSelection(idx, count, d_p, d_n, mean_diff, loss, t["amb_count"], cap)


def mine_semihard_negatives(anchors: torch.Tensor, positives: torch.Tensor, anchor_labels: torch.Tensor,
                            candidates: torch.Tensor, candidate_labels: torch.Tensor):
    """Cross-GPU hard-negative search (BASELINE.json north_star; no reference counterpart).

    `candidates` / `candidate_labels` are normally the RCCL all-gather of every rank's embeddings and
    speaker ids.  For anchor i returns the index j of the candidate with a different speaker that is
    closest to the anchor among those farther than its positive (semi-hard), else the closest one;
    ties -> lowest j; -1 if no candidate has another speaker.  Also returns the distances d(a_i, x_j)."""
    _require_cuda(anchors, "mine_semihard_negatives")
    eng = get_engine()
    a, p, c = (t.detach().contiguous() for t in (anchors, positives, candidates))
    d_p = eng.pairwise_distance(a, p)
    n, d = a.shape
    idx = torch.empty(n, dtype=torch.int64, device=a.device)
    dist_out = torch.empty(n, dtype=torch.float32, device=a.device)
    ws = torch.empty(eng.lib.raw("ds_mine_workspace_floats")(n, c.shape[0]), dtype=torch.float32, device=a.device)
    eng.lib.call("ds_mine_semihard_f32", eng._p(a), eng._p(d_p), eng._p(anchor_labels.to(torch.int64).contiguous()),
                 eng._p(c), eng._p(candidate_labels.to(torch.int64).contiguous()), eng._p(ws), eng._p(idx),
                 eng._p(dist_out), n, c.shape[0], d, eng._stream(a))
    return idx, dist_out
