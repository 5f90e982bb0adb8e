"""Device-resident feature store and batch assembly.

The reference's per-batch host work -- `read_MFB` (np.load, audio_processing.py:38-42), the random
fixed-length crop of `truncatedinputfromMFB` (:58-74), the `totensor` transpose (:185), three times
per triplet in `DeepSpeakerDataset.__getitem__` (DeepSpeakerDataset_dynamic.py:82-103), single-threaded,
followed by an H2D copy (train_triplet.py:210) -- becomes one gather kernel over features that already
sit in HBM.  WHICH utterances and WHERE to crop stay host decisions with the reference's RNG
semantics (np.random for triplets, python `random` for crops); only indices cross
PCIe.  Crops are produced directly in the layout the model consumes, [B,1,T,64].
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np
import torch

from .model import _require_cuda, get_engine

_engine_override = None


def _eng():
    return _engine_override if _engine_override is not None else get_engine()


class FeatureStore:
    """All utterances' [T_u, F] filterbank matrices concatenated into one resident [sum T_u, F] tensor."""

    def __init__(self, utterances: Sequence[np.ndarray], device="cuda"):
        if not utterances:
            raise ValueError("empty corpus")
        f = utterances[0].shape[1]
        lens = [u.shape[0] for u in utterances]
        self.offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        self.n_feat = f
        self.features = torch.from_numpy(np.ascontiguousarray(np.concatenate(utterances, 0), dtype=np.float32)).to(device)

    def __len__(self):
        return len(self.offsets) - 1

    def length(self, utt: int) -> int:
        return int(self.offsets[utt + 1] - self.offsets[utt])

    def crops(self, utt_idx: Sequence[int], starts: Sequence[int], frames: int) -> torch.Tensor:
        """[B,1,frames,F]: rows starts[b] .. starts[b]+frames of utterance utt_idx[b] (zero padded past its end)."""
        eng = _eng()
        utt = np.asarray(utt_idx, np.int64)
        st = np.asarray(starts, np.int64)
        if (st < 0).any() or (utt < 0).any() or (utt >= len(self)).any():
            raise IndexError("crop outside the corpus")
        dev = self.features.device
        # one pinned staging buffer, copied without blocking the host: a pageable .to(device) waits for everything queued
        # on the stream before it -- the caller could never run ahead of the GPU (embed_variable_length, batches in flight)
        rows = torch.from_numpy(np.stack([self.offsets[utt] + st, self.offsets[utt + 1]]))
        if dev.type == "cuda":
            rows = rows.pin_memory().to(dev, non_blocking=True)
        row_start, row_end = rows[0], rows[1]
        out = torch.empty((len(utt), 1, frames, self.n_feat), dtype=torch.float32, device=dev)
        eng.lib.call("ds_assemble_crops_f32", eng._p(self.features), eng._p(row_start), eng._p(row_end), eng._p(out),
                     len(utt), frames, self.n_feat, eng._stream(out))
        return out

    def triplets(self, anchors, positives, negatives, starts, frames: int):
        """The (a, p, n) batches of one step as ONE [3B,1,frames,F] gather (a | p | n); `starts` is [3, B]."""
        b = len(anchors)
        x = self.crops(np.concatenate([anchors, positives, negatives]), np.asarray(starts).reshape(-1), frames)
        return x[:b], x[b:2 * b], x[2 * b:]
