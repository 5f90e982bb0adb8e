"""Triplet selection on the device.

`select_triplets` is the reference's per-triplet filter (train_triplet.py:249-274): keep triplet i
iff d(a_i,n_i) - d(a_i,p_i) < margin, indices ascending as `np.where` returns them.  The reference
round-trips embeddings AND raw inputs through NumPy on the host to do this; here the distances,
the mask and the ordered compaction are kernels and only the selected-row count crosses PCIe.
"""
from __future__ import annotations

import torch

from .model import _require_cuda, get_engine


class TripletSelection:
    """Result of `select_triplets`.  Everything stays on the device; reading `.indices` / `.n_selected` /
    `.n_correct` is what synchronises (the reference branches on the count, train_triplet.py:263)."""

    def __init__(self, idx_full, count, d_p, d_n, mean_diff):
        self._idx_full = idx_full    # int64 [N]; the first `count` entries are valid, ascending
        self.count = count           # int32 [1] on the device
        self.d_p = d_p               # [N]  (train_triplet.py:251)
        self.d_n = d_n               # [N]  (train_triplet.py:252)
        self.mean_diff = mean_diff   # mean(d_n - d_p), 1-element device tensor (train_triplet.py:259-260)

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


def select_triplets(out_a: torch.Tensor, out_p: torch.Tensor, out_n: torch.Tensor, margin: float) -> TripletSelection:
    _require_cuda(out_a, "select_triplets")
    eng = get_engine()
    a, p, n = (t.detach().contiguous() for t in (out_a, out_p, out_n))
    _, d_p, d_n = eng.triplet_margin(a, p, n, margin)
    idx, count, mean_diff = eng.triplet_filter(d_p, d_n, margin)
    return TripletSelection(idx, count, d_p, d_n, mean_diff)


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
