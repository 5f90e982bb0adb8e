"""Verification scoring on the device.

`trial_scores` = the reference's test-time score (train_triplet.py:337-350): both utterances of a trial
are embedded as `crops` fixed-length crops, the distance is taken crop-by-crop and averaged.
`evaluate` = the threshold sweep of eval_metrics.py:5-50 (tpr / fpr / accuracy at the best-accuracy
threshold) plus the equal error rate the reference never computes.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import torch

from .model import _require_cuda, get_engine

_engine_override = None        # tests may bind the host emulator


def _eng():
    return _engine_override if _engine_override is not None else get_engine()


def trial_scores(emb_a: torch.Tensor, emb_p: torch.Tensor, crops: int) -> torch.Tensor:
    """[n_trials*crops, D] x 2 (rows ordered trial-major, crop-minor as train_triplet.py:339-340 builds them)
    -> [n_trials] mean crop-pair distance."""
    eng = _eng()
    d = eng.pairwise_distance(emb_a.contiguous(), emb_p.contiguous())
    n = d.numel() // crops
    out = torch.empty(n, dtype=torch.float32, device=d.device)
    eng.lib.call("ds_group_mean_f32", eng._p(d), eng._p(out), n, crops, eng._stream(d))
    return out


def enrolment_scores(test_emb: torch.Tensor, enrol_emb: torch.Tensor, enrol_sizes) -> torch.Tensor:
    """Score of each trial against its claimed speaker's enrolment set -- sets of different sizes, utterances of
    different lengths (BASELINE configs[4]).  Trial i compares test_emb[i] with the `enrol_sizes[i]` consecutive rows of
    `enrol_emb` that make up its speaker's set and takes the MEAN of the distances, the reference's length
    normalisation (train_triplet.py:348-350 averages a trial's crop-pair distances).  Returns [n_trials]."""
    eng = _eng()
    sizes = np.asarray(enrol_sizes, np.int64)
    if sizes.ndim != 1 or len(sizes) != test_emb.shape[0] or (sizes < 1).any() or int(sizes.sum()) != enrol_emb.shape[0]:
        raise ValueError("enrol_sizes must give one set size >= 1 per trial, summing to the rows of enrol_emb")
    dev = test_emb.device
    def to_dev(a):         # through pinned memory, without blocking the host (a pageable copy waits for the stream's queue)
        t = torch.from_numpy(a)
        return t.pin_memory().to(dev, non_blocking=True) if dev.type == "cuda" else t.to(dev)
    offsets = to_dev(np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64))
    owner = to_dev(np.repeat(np.arange(len(sizes)), sizes).astype(np.int64))
    n, d = enrol_emb.shape
    expanded = torch.empty((n, d), dtype=torch.float32, device=dev)
    eng.lib.call("ds_gather_rows_f32", eng._p(test_emb.contiguous()), eng._p(owner), eng._p(expanded), n, d,
                 eng._stream(expanded))
    dist = eng.pairwise_distance(expanded, enrol_emb.contiguous())
    out = torch.empty(len(sizes), dtype=torch.float32, device=dev)
    eng.lib.call("ds_segment_mean_f32", eng._p(dist), eng._p(offsets), eng._p(out), len(sizes), eng._stream(dist))
    return out


@dataclass
class Verification:
    tpr: float
    fpr: float
    accuracy: float
    threshold: float
    eer: float
    eer_threshold: float
    tp: torch.Tensor        # per-threshold counts, on the device
    fp: torch.Tensor


def evaluate(distances: torch.Tensor, labels: torch.Tensor, thr_start: float = 0.0, thr_stop: float = 30.0,
             thr_step: float = 0.01) -> Verification:
    """eval_metrics.evaluate (thresholds np.arange(0, 30, 0.01), eval_metrics.py:7) on the device + EER."""
    eng = _eng()
    d = distances.contiguous().float()
    lab = (labels != 0).to(torch.int32).contiguous()
    n = d.numel()
    n_thr = len(np.arange(thr_start, thr_stop, thr_step))
    n_same = int(lab.sum().item())
    tp = torch.empty(n_thr, dtype=torch.int32, device=d.device)
    fp = torch.empty_like(tp)
    summary = torch.empty(6, dtype=torch.float32, device=d.device)
    eng.lib.call("ds_roc_sweep_f32", eng._p(d), eng._p(lab), n, float(thr_start), float(thr_step), n_thr, n_same,
                 n - n_same, eng._p(tp), eng._p(fp), eng._p(summary), eng._stream(d))
    s = summary.cpu().numpy()
    return Verification(float(s[1]), float(s[2]), float(s[3]), thr_start + thr_step * float(s[0]), float(s[4]),
                        float(s[5]), tp, fp)
