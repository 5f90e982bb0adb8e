"""Host-side orchestration of the HIP kernels for the Deep Speaker ResCNN.

`Engine` turns one call of the reference's `DeepSpeakerModel.forward` (reference model.py:185-218)
into the sequence of C-ABI launches declared in include/deepspeaker_hip.h.  It owns no arithmetic:
every tensor op is a kernel in libdeepspeaker_hip.so; torch supplies buffers and the stream.

Data layout in HBM: activations are channels-last fp32 `[B, T', F', C]`; the network
input `[B,1,T,64]` is consumed in place (C = 1).  Filters are packed once per weight version.
"""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch

from ._native import (ConvShape, DS_CONV_IN_PLANES16, DS_EPI_AFFINE, DS_EPI_CLIP, DS_EPI_OUT_F16, DS_EPI_OUT_F32,
                      DS_EPI_OUT_PLANES16, DS_EPI_RESIDUAL, DS_EPI_STATS, DS_TAIL_SMALL_MAX_B, NativeLib, PackJob)

PACK_BATCH_MAX = 32                         # DS_PACK_BATCH_MAX (include/deepspeaker_hip.h)

PRECISIONS = ("f32", "bf16x3", "bf16", "f16")

STAGE_CHANNELS = (64, 128, 256, 512)        # reference model.py:93-107
BN_EPS = 1e-5
BN_MOMENTUM = 0.1
L2_EPS = 1e-10                               # reference model.py:176
ALPHA = 10.0                                 # reference model.py:212


def conv_out(n: int, k: int, stride: int) -> int:
    return (n + 2 * (k // 2) - k) // stride + 1


@dataclass
class BNParams:
    weight: torch.Tensor
    bias: torch.Tensor
    running_mean: torch.Tensor
    running_var: torch.Tensor


@dataclass
class StageWeights:
    conv: torch.Tensor              # packed 5x5 (stage 1: [25][64])
    l_conv1: torch.Tensor           # packed 3x3
    l_conv2: torch.Tensor
    conv_dgrad: Optional[torch.Tensor] = None
    l_conv1_dgrad: Optional[torch.Tensor] = None
    l_conv2_dgrad: Optional[torch.Tensor] = None
    # bf16 matrix-core banks: (hi, lo) pairs of [Cin/16][tap][Cout][16] bf16 (precision "bf16x3" / "bf16")
    conv_bf16: Optional[Tuple[torch.Tensor, torch.Tensor]] = None
    l_conv1_bf16: Optional[Tuple[torch.Tensor, torch.Tensor]] = None
    l_conv2_bf16: Optional[Tuple[torch.Tensor, torch.Tensor]] = None
    l_conv1_dgrad_bf16: Optional[Tuple[torch.Tensor, torch.Tensor]] = None
    l_conv2_dgrad_bf16: Optional[Tuple[torch.Tensor, torch.Tensor]] = None
    conv_dgrad_bf16: Optional[Tuple[torch.Tensor, torch.Tensor]] = None     # 5x5 stride 2: four parity-class banks
    # fp16 matrix-core banks [Cin/16][tap][Cout][16] (precision "f16")
    conv_f16: Optional[torch.Tensor] = None
    l_conv1_f16: Optional[torch.Tensor] = None
    l_conv2_f16: Optional[torch.Tensor] = None
    # ... and the data-gradient banks of the fp16 training step (train_f16.py): flipped 3x3 banks; the 5x5 stride-2 layer's
    # four parity classes as one 3x3 bank with 4 Cin output channels
    conv_dgrad_f16: Optional[torch.Tensor] = None
    l_conv1_dgrad_f16: Optional[torch.Tensor] = None
    l_conv2_dgrad_f16: Optional[torch.Tensor] = None


@dataclass
class PackedWeights:
    stages: List[StageWeights]
    fc: torch.Tensor                # packed as a 1x1 convolution over k' = f*C + c
    fc_bias: torch.Tensor
    fc_ones: torch.Tensor
    fc_dgrad: Optional[torch.Tensor] = None
    fc_rows: Optional[torch.Tensor] = None      # [N][K'] row-major copy for the small-batch tail (built on first use)
    fc_src: Optional[torch.Tensor] = None       # the fc.weight the copies were made from, and its (C, F) split
    fc_cf: Optional[Tuple[int, int]] = None
    owner: Optional[int] = None     # id of the model these were packed for (plan-cache generations), or None


@dataclass
class SavedForward:
    """What the backward pass needs from one train-mode forward (all channels-last)."""
    x: torch.Tensor
    acts: Dict[str, torch.Tensor] = field(default_factory=dict)      # post-activation tensors
    raws: Dict[str, torch.Tensor] = field(default_factory=dict)      # raw conv outputs (BN inputs)
    stats: Dict[str, Tuple[torch.Tensor, ...]] = field(default_factory=dict)     # (mean, invstd, scale, shift), or a list of them per member
    pooled: Optional[torch.Tensor] = None
    fc_out: Optional[torch.Tensor] = None
    dims: List[Tuple[int, int]] = field(default_factory=list)


# entry points that launch exactly one (big-LDS, MFMA) kernel: profiled through events bound to the launch itself
_SELF_TIMED = frozenset({"ds_conv_fwd_f16", "ds_conv_block_f16", "ds_conv_block_f16_masked"})


class LaunchEvent:
    """A HIP event of ours (ds_event_create) for `ds_launch_timing_arm`: a pair bound to a launch reads that kernel's
    execution time, with the `elapsed_time` of torch.cuda.Event (milliseconds; waits for the second event)."""

    def __init__(self, lib):
        self._lib = lib
        h = ctypes.c_void_p()
        lib.call("ds_event_create", ctypes.byref(h))
        self.handle = h

    def elapsed_time(self, other: "LaunchEvent") -> float:
        ms = ctypes.c_float()
        self._lib.call("ds_event_elapsed_ms", self.handle, other.handle, ctypes.byref(ms))
        return float(ms.value)

    def __del__(self):
        try:
            if self.handle:
                self._lib.raw("ds_event_destroy")(self.handle)
        except Exception:
            pass


class _Range:
    """roctx range (SURVEY section 5 tracing): `DS_ROCTX=1` brackets the phases of a step -- eval forward, each timed
    convolution launch, train forward / backward / optimizer -- with roctxRangePush / Pop (torch.cuda.nvtx is roctx on
    ROCm), so that `rocprofv3 --marker-trace` shows them next to the kernels.  Off by default: a no-op object."""
    enabled = os.environ.get("DS_ROCTX", "0") not in ("", "0")

    def __init__(self, name: str):
        self.name = name

    def __enter__(self):
        if _Range.enabled:
            torch.cuda.nvtx.range_push(self.name)
        return self

    def __exit__(self, *exc):
        if _Range.enabled:
            torch.cuda.nvtx.range_pop()
        return False


def trace_range(name: str) -> _Range:
    return _Range(name)


class Engine:
    PLAN_CACHE_ENTRIES = 64            # eval launch plans kept (each owns its activation buffers), LRU
    PLAN_CACHE_BYTES = 12 << 30        # ... and their total size (a 768 x 160-frame f32-class plan is 3.3 GiB)

    def __init__(self, lib: NativeLib):
        self.lib = lib
        # When set to a list, every implicit-GEMM convolution launch is bracketed by two events on the
        # launch stream and (label, flops, start, end, arithmetic) is appended (bench.py's live roofline).
        self.profile: Optional[list] = None
        self.self_timed_launches = True     # profile entries of single-kernel calls from events bound to the launch
        self.profile_every = 1              # planned forwards: time the launches of every N-th call only (a timed launch
        self._profile_calls = {}            # costs ~5 us of completion-signal handling: 2 % of the eval step at N = 1);
                                            # counted per precision (a refinement forward does not shift the main one's turn)

    # ------------------------------------------------------------------ plumbing
    @staticmethod
    def _stream(t: torch.Tensor):
        if t.is_cuda:
            return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
        return None

    @staticmethod
    def _p(t: Optional[torch.Tensor]):
        return None if t is None else ctypes.c_void_p(t.data_ptr())

    @staticmethod
    def _check(t: torch.Tensor, name: str):
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise ValueError(f"{name}: expected a contiguous float32 tensor, got {t.dtype}, "
                             f"contiguous={t.is_contiguous()}")

    # ------------------------------------------------------------------ weights
    # Filters are packed in batches: the helpers below allocate the bank and QUEUE a job; `_flush_packs` runs every queued
    # job of a family in one launch (ds_pack_conv_weights_{f16,bf16}_batch).  A training step re-packs every filter after
    # every optimizer step: 22 (fp16) / 19 (bf16x3) launches of ~5 us each were 0.17 ms at the head of every step.
    def _pack_bf16(self, w: torch.Tensor, ks: int, dgrad: bool = False, jobs: Optional[list] = None):
        hi = torch.empty(w.numel(), dtype=torch.bfloat16, device=w.device)
        lo = torch.empty_like(hi)
        job = (w, hi, lo, w.shape[0], w.shape[1], ks, 1 if dgrad else 0)
        if jobs is None:
            self._flush_packs("bf16", [job])
        else:
            jobs.append(job)
        return hi, lo

    def _pack_f16(self, w: torch.Tensor, ks: int, jobs: Optional[list] = None):
        out = torch.empty(w.numel(), dtype=torch.float16, device=w.device)
        job = (w, out, None, w.shape[0], w.shape[1], ks, 0)
        if jobs is None:
            self._flush_packs("f16", [job])
        else:
            jobs.append(job)
        return out

    def _pack_f16_dgrad(self, w: torch.Tensor, ks: int, stride: int, jobs: Optional[list] = None):
        n = w.shape[0] * w.shape[1] * (36 if stride == 2 else ks * ks)
        out = torch.empty(n, dtype=torch.float16, device=w.device)
        job = (w, out, None, w.shape[0], w.shape[1], ks, 2 if stride == 2 else 1)
        if jobs is None:
            self._flush_packs("f16", [job])
        else:
            jobs.append(job)
        return out

    def _flush_packs(self, family: str, jobs: list):
        """one launch per DS_PACK_BATCH_MAX queued jobs of a family ("f16" / "bf16")"""
        name = "ds_pack_conv_weights_f16_batch" if family == "f16" else "ds_pack_conv_weights_bf16_batch"
        for k in range(0, len(jobs), PACK_BATCH_MAX):
            chunk = jobs[k:k + PACK_BATCH_MAX]
            arr = (PackJob * len(chunk))()
            for a, (w, out, out2, co, ci, ks, mode) in zip(arr, chunk):
                self._check(w, "filter")
                a.w_oihw, a.out = w.data_ptr(), out.data_ptr()
                a.out2 = out2.data_ptr() if out2 is not None else None
                a.Cout, a.Cin, a.KS, a.mode = co, ci, ks, mode
            self.lib.call(name, arr, len(chunk), self._stream(chunk[0][0]))

    def pack_weights(self, sd: Dict[str, torch.Tensor], n_stages: int = 4,
                     with_dgrad: bool = False, with_bf16: bool = False, with_f16: bool = False,
                     f32_banks: bool = True, with_f16_dgrad: bool = False) -> PackedWeights:
        """OIHW / [out,in] parameters (reference shapes) -> kernel layouts.  `f32_banks=False`
        (with_bf16): the f32 banks of the 3x3 / 5x5 layers are not built -- a bf16x3 training step re-packs every
        filter after every optimizer step and never reads them (21 launches per step)."""
        if not f32_banks and not (with_bf16 or (with_f16 and with_f16_dgrad)):
            raise ValueError("f32_banks=False needs with_bf16=True (or the fp16 training banks)")
        lib = self.lib
        stages = []
        jobs16, jobsb = [], []
        for s in range(n_stages):
            i = s + 1
            w = sd[f"model.conv{i}.weight"].detach()
            self._check(w, f"model.conv{i}.weight")
            st = self._stream(w)
            co, ci = w.shape[0], w.shape[1]
            pc = torch.empty(w.numel(), dtype=torch.float32, device=w.device) if (i == 1 or f32_banks) else None
            if i == 1:
                lib.call("ds_pack_conv1_weight_f32", self._p(w), self._p(pc), co, st)
            elif f32_banks:
                lib.call("ds_pack_conv_weight_f32", self._p(w), self._p(pc), co, ci, 5, 0, st)
            packs = []
            for j in (1, 2):
                wl = sd[f"model.layer{i}.0.conv{j}.weight"].detach()
                self._check(wl, f"model.layer{i}.0.conv{j}.weight")
                pl = None
                if f32_banks:
                    pl = torch.empty(wl.numel(), dtype=torch.float32, device=wl.device)
                    lib.call("ds_pack_conv_weight_f32", self._p(wl), self._p(pl), co, co, 3, 0, st)
                packs.append(pl)
            sw = StageWeights(pc, packs[0], packs[1])
            if with_f16:
                if i > 1:
                    sw.conv_f16 = self._pack_f16(w, 5, jobs16)
                sw.l_conv1_f16 = self._pack_f16(sd[f"model.layer{i}.0.conv1.weight"].detach(), 3, jobs16)
                sw.l_conv2_f16 = self._pack_f16(sd[f"model.layer{i}.0.conv2.weight"].detach(), 3, jobs16)
                if with_f16_dgrad:
                    if i > 1:
                        sw.conv_dgrad_f16 = self._pack_f16_dgrad(w, 5, 2, jobs16)
                    sw.l_conv1_dgrad_f16 = self._pack_f16_dgrad(sd[f"model.layer{i}.0.conv1.weight"].detach(), 3, 1, jobs16)
                    sw.l_conv2_dgrad_f16 = self._pack_f16_dgrad(sd[f"model.layer{i}.0.conv2.weight"].detach(), 3, 1, jobs16)
            if with_bf16:
                if i > 1:
                    sw.conv_bf16 = self._pack_bf16(w, 5, jobs=jobsb)
                sw.l_conv1_bf16 = self._pack_bf16(sd[f"model.layer{i}.0.conv1.weight"].detach(), 3, jobs=jobsb)
                sw.l_conv2_bf16 = self._pack_bf16(sd[f"model.layer{i}.0.conv2.weight"].detach(), 3, jobs=jobsb)
                if with_dgrad:
                    if i > 1:
                        hi = torch.empty(36 * co * ci, dtype=torch.bfloat16, device=w.device)
                        lo = torch.empty_like(hi)
                        lib.call("ds_pack_conv_weight_dgrad_s2_bf16", self._p(w), self._p(hi), self._p(lo), co, ci, st)
                        sw.conv_dgrad_bf16 = (hi, lo)
                    sw.l_conv1_dgrad_bf16 = self._pack_bf16(sd[f"model.layer{i}.0.conv1.weight"].detach(), 3, True, jobsb)
                    sw.l_conv2_dgrad_bf16 = self._pack_bf16(sd[f"model.layer{i}.0.conv2.weight"].detach(), 3, True, jobsb)
            if with_dgrad and f32_banks:
                if i > 1:       # 5x5 stride 2: four parity-class banks (ds_conv_dgrad_f32)
                    sw.conv_dgrad = torch.empty_like(pc)
                    lib.call("ds_pack_conv_dgrad_s2_f32", self._p(w), self._p(sw.conv_dgrad), co, ci, st)
                for j, attr in ((1, "l_conv1_dgrad"), (2, "l_conv2_dgrad")):
                    wl = sd[f"model.layer{i}.0.conv{j}.weight"].detach()
                    pd = torch.empty(wl.numel(), dtype=torch.float32, device=wl.device)
                    lib.call("ds_pack_conv_weight_f32", self._p(wl), self._p(pd), co, co, 3, 1, st)
                    setattr(sw, attr, pd)
            stages.append(sw)
        if jobs16:
            self._flush_packs("f16", jobs16)
        if jobsb:
            self._flush_packs("bf16", jobsb)
        wfc = sd["model.fc.weight"].detach()
        bfc = sd["model.fc.bias"].detach()
        self._check(wfc, "model.fc.weight")
        c_last = STAGE_CHANNELS[n_stages - 1]
        f_bins = wfc.shape[1] // c_last
        pfc = torch.empty(wfc.numel(), dtype=torch.float32, device=wfc.device)
        lib.call("ds_pack_fc_weight_f32", self._p(wfc), self._p(pfc), wfc.shape[0], c_last, f_bins,
                 self._stream(wfc))
        ones = torch.ones(wfc.shape[0], dtype=torch.float32, device=wfc.device)
        pw = PackedWeights(stages, pfc, bfc.contiguous(), ones)
        pw.fc_src, pw.fc_cf = wfc, (c_last, f_bins)
        if with_dgrad:
            pw.fc_dgrad = torch.empty_like(pfc)
            lib.call("ds_pack_fc_weight_dgrad_f32", self._p(wfc), self._p(pw.fc_dgrad), wfc.shape[0], c_last,
                     f_bins, self._stream(wfc))
        return pw

    def bn_fold(self, bn: BNParams) -> Tuple[torch.Tensor, torch.Tensor]:
        c = bn.weight.numel()
        scale = torch.empty(c, dtype=torch.float32, device=bn.weight.device)
        shift = torch.empty_like(scale)
        self.lib.call("ds_bn_fold_f32", self._p(bn.weight.detach()), self._p(bn.bias.detach()),
                      self._p(bn.running_mean), self._p(bn.running_var), BN_EPS, self._p(scale),
                      self._p(shift), c, self._stream(scale))
        return scale, shift

    # ------------------------------------------------------------------ single launches
    def conv(self, x: torch.Tensor, wp: torch.Tensor, B: int, H: int, W: int, cin: int, cout: int,
             ks: int, stride: int, scale=None, shift=None, residual=None, flags: int = 0,
             want_stats: bool = False, out=None):
        shp = ConvShape(B, H, W, cin, cout, ks, stride)
        ho, wo = conv_out(H, ks, stride), conv_out(W, ks, stride)
        y = out if out is not None else torch.empty((B, ho, wo, cout), dtype=torch.float32, device=x.device)
        stats = None
        if want_stats:
            rows = self.lib.raw("ds_conv_stats_rows")(ctypes.byref(shp))
            if rows <= 0:
                raise RuntimeError(f"ds_conv_stats_rows failed: {rows}")
            stats = torch.empty((rows, cout, 2), dtype=torch.float32, device=x.device)
            flags |= DS_EPI_STATS
        prof = self.profile is not None and x.is_cuda
        if prof:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        self.lib.call("ds_conv_fwd_f32", ctypes.byref(shp), self._p(x), self._p(wp), self._p(scale),
                      self._p(shift), self._p(residual), self._p(y), self._p(stats), flags, self._stream(x))
        if prof:
            ev1.record()
            self.profile.append((f"conv{ks}x{ks}s{stride}_{cin}to{cout}_{ho}x{wo}",
                                 2.0 * B * ho * wo * cout * cin * ks * ks, ev0, ev1, "f32"))
        return y, stats

    def conv_bf16(self, x: torch.Tensor, w_pair, x3: bool, B: int, H: int, W: int, cin: int, cout: int, ks: int,
                  stride: int, scale=None, shift=None, residual=None, flags: int = 0, want_stats: bool = False,
                  out=None):
        """Forward convolution on the bf16 matrix cores; x3 = hi/lo split operands (f32-class accuracy)."""
        shp = ConvShape(B, H, W, cin, cout, ks, stride)
        ho, wo = conv_out(H, ks, stride), conv_out(W, ks, stride)
        y = out if out is not None else torch.empty((B, ho, wo, cout), dtype=torch.float32, device=x.device)
        stats = None
        if want_stats:
            rows = self.lib.raw("ds_conv_bf16_stats_rows")(ctypes.byref(shp), int(x3))
            if rows <= 0:
                raise RuntimeError(f"ds_conv_bf16_stats_rows failed: {rows}")
            stats = torch.empty((rows, cout, 2), dtype=torch.float32, device=x.device)
            flags |= DS_EPI_STATS
        prof = self.profile is not None and x.is_cuda
        if prof:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        self.lib.call("ds_conv_fwd_bf16", ctypes.byref(shp), self._p(x), self._p(w_pair[0]),
                      self._p(w_pair[1]) if x3 else None, self._p(scale), self._p(shift), self._p(residual),
                      self._p(y), self._p(stats), flags, self._stream(x))
        if prof:
            ev1.record()
            self.profile.append((f"conv{ks}x{ks}s{stride}_{cin}to{cout}_{ho}x{wo}",
                                 2.0 * B * ho * wo * cout * cin * ks * ks, ev0, ev1, "bf16x3" if x3 else "bf16"))
        return (y, stats) if want_stats else y

    def conv_f16(self, x: torch.Tensor, w_f16: torch.Tensor, B: int, H: int, W: int, cin: int, cout: int, ks: int,
                 stride: int, scale=None, shift=None, residual=None, flags: int = 0, out=None):
        """Forward convolution on the fp16 matrix cores: fp16 channels-last in, fp16 out (f32 with DS_EPI_OUT_F32)."""
        shp = ConvShape(B, H, W, cin, cout, ks, stride)
        ho, wo = conv_out(H, ks, stride), conv_out(W, ks, stride)
        y = out if out is not None else torch.empty((B, ho, wo, cout), dtype=torch.float32 if flags & DS_EPI_OUT_F32
                                                    else torch.float16, device=x.device)
        prof = self.profile is not None and x.is_cuda
        if prof:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        self.lib.call("ds_conv_fwd_f16", ctypes.byref(shp), self._p(x), self._p(w_f16), self._p(scale), self._p(shift),
                      self._p(residual), self._p(y), flags, self._stream(x))
        if prof:
            ev1.record()
            self.profile.append((f"conv{ks}x{ks}s{stride}_{cin}to{cout}_{ho}x{wo}",
                                 2.0 * B * ho * wo * cout * cin * ks * ks, ev0, ev1, "f16"))
        return y

    def conv1(self, x: torch.Tensor, wp: torch.Tensor, B: int, H: int, W: int, scale=None, shift=None,
              flags: int = 0, want_stats: bool = False, lowp: bool = False, out=None):
        ho, wo = conv_out(H, 5, 2), conv_out(W, 5, 2)
        y = out if out is not None else torch.empty((B, ho, wo, 64), dtype=torch.float16 if flags & DS_EPI_OUT_F16
                                                    else torch.float32, device=x.device)
        stats = None
        if want_stats:
            rows = self.lib.raw("ds_conv5x5s2_c1_stats_rows")(B, H)
            stats = torch.empty((rows, 64, 2), dtype=torch.float32, device=x.device)
            flags |= DS_EPI_STATS
        # lowp: the split-operand bf16 matrix-core kernel (same bank, same contract); f32: the exact VALU kernel
        self.lib.call("ds_conv5x5s2_c1_fwd_bf16" if lowp else "ds_conv5x5s2_c1_fwd_f32", self._p(x), self._p(wp),
                      self._p(scale), self._p(shift), self._p(y), self._p(stats), B, H, W, 64, flags, self._stream(x))
        return y, stats

    def bn_finalize(self, stats: torch.Tensor, count: int, bn: BNParams, update_running: bool = True,
                    reducer=None, out=None):
        """Per-tile partial sums -> batch mean / invstd / (scale, shift) + running-stat update.  With a
        `reducer` (data-parallel training) the [C][2] float64 sums and the pixel count are summed over
        the ranks first, so every rank normalises with the statistics of the global batch."""
        c = bn.weight.numel()
        dev = stats.device
        if out is not None:                     # (mean, invstd[, scale, shift]) destinations, e.g. rows of per-member tables
            mean, invstd = out[0], out[1]
        else:
            mean = torch.empty(c, dtype=torch.float32, device=dev)
            invstd = torch.empty_like(mean)
        if out is not None and len(out) == 4:
            scale, shift = out[2], out[3]
        else:
            scale = torch.empty(c, dtype=torch.float32, device=dev)
            shift = torch.empty_like(scale)
        if reducer is not None and reducer.active:
            sums = torch.empty(2 * c + 1, dtype=torch.float64, device=dev)
            self.lib.call("ds_partial_sum_f64", self._p(stats), stats.shape[0], self._p(sums), c, self._stream(stats))
            sums[2 * c] = float(count)
            reducer.all_reduce_sum_(sums)
            self.lib.call("ds_bn_stats_from_sums_f32", self._p(sums), 0, self._p(bn.weight.detach()),
                          self._p(bn.bias.detach()), BN_EPS, BN_MOMENTUM,
                          self._p(bn.running_mean) if update_running else None,
                          self._p(bn.running_var) if update_running else None,
                          self._p(mean), self._p(invstd), self._p(scale), self._p(shift), c, self._stream(stats))
            return mean, invstd, scale, shift
        self.lib.call("ds_bn_stats_finalize_f32", self._p(stats), stats.shape[0], count,
                      self._p(bn.weight.detach()), self._p(bn.bias.detach()), BN_EPS, BN_MOMENTUM,
                      self._p(bn.running_mean) if update_running else None,
                      self._p(bn.running_var) if update_running else None,
                      self._p(mean), self._p(invstd), self._p(scale), self._p(shift), c, self._stream(stats))
        return mean, invstd, scale, shift

    def bn_apply(self, x: torch.Tensor, scale, shift, residual=None, flags: int = 0, out=None):
        y = out if out is not None else torch.empty_like(x)
        c = x.shape[-1]
        self.lib.call("ds_bn_apply_f32", self._p(x), self._p(scale), self._p(shift), self._p(residual),
                      self._p(y), x.numel() // c, c, flags, self._stream(x))
        return y

    # ------------------------------------------------------------------ tail
    def tail(self, a: torch.Tensor, pw: PackedWeights, saved: Optional[SavedForward] = None):
        B, hr, wc, c = a.shape
        pooled = torch.empty((B, wc * c), dtype=torch.float32, device=a.device)
        self.lib.call("ds_avgpool_time_f32", self._p(a), self._p(pooled), B, hr, wc, c, self._stream(a))
        n_out = pw.fc_bias.numel()
        k = wc * c
        ws_floats = self.lib.raw("ds_fc_workspace_floats")(B, k, n_out)
        if ws_floats <= 0:
            raise RuntimeError(f"ds_fc_workspace_floats({B},{k},{n_out}) failed: {ws_floats}")
        ws = torch.empty(ws_floats, dtype=torch.float32, device=a.device)
        f = torch.empty((B, n_out), dtype=torch.float32, device=a.device)
        e = torch.empty_like(f)
        # projection (reference model.py:209) + L2 norm x alpha (model.py:210-213) in two launches
        self.lib.call("ds_fc_l2norm_fwd_f32", self._p(pooled), self._p(pw.fc), self._p(pw.fc_bias.detach()),
                      self._p(ws), self._p(f), self._p(e), B, k, n_out, ALPHA, L2_EPS, self._stream(a))
        if saved is not None:
            saved.pooled, saved.fc_out = pooled, f
        return e

    # ------------------------------------------------------------------ cached launch plan (eval)
    def _build_eval_plan(self, x, pw: PackedWeights, folded, precision: str, masked: bool = False,
                         low_latency: bool = False):
        """Everything that does not change between two eval forwards of the same shape -- tile plans are
        recomputed inside the library anyway, but the Python side of a launch (shape structs, pointer
        objects, activation buffers, the stream handle lookup) costs more than the launch itself at
        ~100k embeddings/s.  The plan owns the intermediate activations (re-used by the next call on the
        same stream); input, output and stream are patched into three mutable ctypes slots per call."""
        B, _, T, F = x.shape
        dev = x.device
        lowp = precision != "f32"
        x3 = precision == "bf16x3"
        h16 = precision == "f16"
        x_slot, e_slot, st_slot = ctypes.c_void_p(0), ctypes.c_void_p(0), ctypes.c_void_p(0)
        calls = []            # (raw function, argument tuple, profile label or None, flops)
        keep = []             # tensors / structs the argument tuples point into
        AC = DS_EPI_AFFINE | DS_EPI_CLIP
        # variable-length batch: per stage the number of rows each utterance really has; after every layer the rows
        # past it are re-zeroed (ds_mask_rows) so that they keep acting as that utterance's zero padding
        lens_dev = torch.zeros((len(pw.stages), B), dtype=torch.int32, device=dev) if masked else None

        def mask_call(t, stage, hh):
            if masked:
                row_bytes = t.shape[2] * t.shape[3] * t.element_size()
                calls.append((self.lib.raw("ds_mask_rows"), (self._p(t), self._p(lens_dev[stage]), B, hh, row_bytes, st_slot),
                              None, 0.0))

        def buf(*shape, dtype=torch.float32):
            t = torch.empty(shape, dtype=dtype, device=dev)
            keep.append(t)
            return t

        def conv_call(src_p, w_f32, w_bf16, Bc, h, w, cin, cout, ks, stride, sc, sh, res, w_f16=None, last=False,
                      layout_flags=0):
            shp = ConvShape(Bc, h, w, cin, cout, ks, stride)
            keep.append(shp)
            ho, wo = conv_out(h, ks, stride), conv_out(w, ks, stride)
            flags = AC | (DS_EPI_RESIDUAL if res is not None else 0) | layout_flags
            label = f"conv{ks}x{ks}s{stride}_{cin}to{cout}_{ho}x{wo}"
            flops = 2.0 * Bc * ho * wo * cout * cin * ks * ks
            if h16:         # fp16 activations between the layers; the last layer hands f32 to the tail
                y = buf(Bc, ho, wo, cout, dtype=torch.float32 if last else torch.float16)
                ws_bytes = self.lib.raw("ds_conv_f16_splitk_workspace_bytes")(ctypes.byref(shp)) if low_latency else 0
                if ws_bytes > 0:        # a launch too small to fill the GPU: contraction split over workgroups
                    ws = buf(ws_bytes // 4)
                    args = (ctypes.byref(shp), src_p, self._p(w_f16), self._p(sc), self._p(sh), self._p(res), self._p(y),
                            flags | (DS_EPI_OUT_F32 if last else 0), self._p(ws), ws_bytes, st_slot)
                    calls.append((self.lib.raw("ds_conv_fwd_f16_splitk"), args, label, flops))
                    return y, ho, wo
                args = (ctypes.byref(shp), src_p, self._p(w_f16), self._p(sc), self._p(sh), self._p(res), self._p(y),
                        flags | (DS_EPI_OUT_F32 if last else 0), st_slot)
                calls.append((self.lib.raw("ds_conv_fwd_f16"), args, label, flops))
                return y, ho, wo
            y = buf(Bc, ho, wo, cout)
            if lowp:
                args = (ctypes.byref(shp), src_p, self._p(w_bf16[0]), self._p(w_bf16[1]) if x3 else None,
                        self._p(sc), self._p(sh), self._p(res), self._p(y), None, flags, st_slot)
                calls.append((self.lib.raw("ds_conv_fwd_bf16"), args, label, flops))
            else:
                args = (ctypes.byref(shp), src_p, self._p(w_f32), self._p(sc), self._p(sh), self._p(res), self._p(y),
                        None, flags, st_slot)
                calls.append((self.lib.raw("ds_conv_fwd_f32"), args, label, flops))
            return y, ho, wo

        h, w, cin = T, F, 1
        a = None
        # fp16 path: the 64-channel tensor between stage 1 and the 64->128 5x5 layer travels channel-plane-major
        # ([4][pixels][16]): that layer works in 16-channel chunks, and a channels-last 64-channel record is one
        # 128-byte line of which every chunk would read a quarter (4x the HBM / L2 traffic, measured).  Nobody else
        # reads that tensor (masked variable-length plans too: the fused block zeroes the rows past an utterance's
        # extent itself).  The split-K small-launch plans keep channels-last.
        planes = h16 and not low_latency and len(pw.stages) > 1
        for s, sw in enumerate(pw.stages):
            i, c = s + 1, STAGE_CHANNELS[s]
            sc, sh = folded[f"model.bn{i}"]
            if i == 1:
                ho, wo = conv_out(h, 5, 2), conv_out(w, 5, 2)
                a = buf(B, ho, wo, 64, dtype=torch.float16 if h16 else torch.float32)
                calls.append((self.lib.raw("ds_conv5x5s2_c1_fwd_bf16" if precision != "f32" else "ds_conv5x5s2_c1_fwd_f32"),
                              (x_slot, self._p(sw.conv), self._p(sc), self._p(sh), self._p(a), None, B, h, w, 64,
                               AC | (DS_EPI_OUT_F16 if h16 else 0), st_slot), None, 0.0))
                h, w = ho, wo
            else:
                a, h, w = conv_call(self._p(a), sw.conv, sw.conv_bf16, B, h, w, cin, c, 5, 2, sc, sh, None, sw.conv_f16,
                                    layout_flags=DS_CONV_IN_PLANES16 if (planes and i == 2) else 0)
            mask_call(a, s, h)
            cin = c
            last_stage = s == len(pw.stages) - 1
            if h16 and self.lib.raw("ds_conv_block_f16_supported")(B, h, w, c) == 1:
                # the shallow stages: the whole BasicBlock as one kernel, the intermediate activation in LDS only
                # (masked batches: the kernel zeroes the rows past each utterance's extent after both layers itself)
                sc1, sh1 = folded[f"model.layer{i}.0.bn1"]
                sc2, sh2 = folded[f"model.layer{i}.0.bn2"]
                out = buf(B, h, w, c, dtype=torch.float32 if last_stage else torch.float16)
                fl = (DS_EPI_OUT_F32 if last_stage else 0) | (DS_EPI_OUT_PLANES16 if (planes and i == 1) else 0)
                if masked:
                    calls.append((self.lib.raw("ds_conv_block_f16_masked"),
                                  (self._p(a), self._p(sw.l_conv1_f16), self._p(sw.l_conv2_f16), self._p(sc1), self._p(sh1),
                                   self._p(sc2), self._p(sh2), self._p(out), self._p(lens_dev[s]), B, h, w, c, fl, st_slot),
                                  f"block3x3_{c}_{h}x{w}", 2 * 2.0 * B * h * w * c * c * 9))
                else:
                    calls.append((self.lib.raw("ds_conv_block_f16"),
                                  (self._p(a), self._p(sw.l_conv1_f16), self._p(sw.l_conv2_f16), self._p(sc1), self._p(sh1),
                                   self._p(sc2), self._p(sh2), self._p(out), B, h, w, c, fl, st_slot),
                                  f"block3x3_{c}_{h}x{w}", 2 * 2.0 * B * h * w * c * c * 9))
                a = out
                continue
            sc, sh = folded[f"model.layer{i}.0.bn1"]
            y, _, _ = conv_call(self._p(a), sw.l_conv1, sw.l_conv1_bf16, B, h, w, c, c, 3, 1, sc, sh, None, sw.l_conv1_f16)
            mask_call(y, s, h)
            sc, sh = folded[f"model.layer{i}.0.bn2"]
            a, _, _ = conv_call(self._p(y), sw.l_conv2, sw.l_conv2_bf16, B, h, w, c, c, 3, 1, sc, sh, a, sw.l_conv2_f16,
                                last=(s == len(pw.stages) - 1),
                                layout_flags=DS_EPI_OUT_PLANES16 if (planes and i == 1) else 0)
            if s < len(pw.stages) - 1:          # (the masked pool below never reads the last stage's padding rows)
                mask_call(a, s, h)
        k = w * cin
        n_out = pw.fc_bias.numel()
        if low_latency and not masked and B <= DS_TAIL_SMALL_MAX_B and k * 4 <= 65536 and pw.fc_src is not None:
            # serving a few utterances: pooling + projection in one launch, then the norm (8 us instead of 30 at B = 1)
            if pw.fc_rows is None:
                pw.fc_rows = torch.empty(pw.fc_src.numel(), dtype=torch.float32, device=dev)
                self.lib.call("ds_pack_fc_weight_rows_f32", self._p(pw.fc_src), self._p(pw.fc_rows), n_out, pw.fc_cf[0],
                              pw.fc_cf[1], self._stream(x))
                # packed lazily on the stream of the first plan that wants it, then shared: plans built later for OTHER
                # streams (BatchesInFlight lanes, the refinement's side stream) order themselves after this event
                pw.fc_rows_event = torch.cuda.current_stream(dev).record_event() if x.is_cuda else None
            elif x.is_cuda and getattr(pw, "fc_rows_event", None) is not None:
                torch.cuda.current_stream(dev).wait_event(pw.fc_rows_event)
            f = buf(B, n_out)
            calls.append((self.lib.raw("ds_tail_small_f32"),
                          (self._p(a), self._p(pw.fc_rows), self._p(pw.fc_bias.detach()), self._p(f), e_slot, B, h, k, n_out,
                           ALPHA, L2_EPS, st_slot), None, 0.0))
            keep += [pw, folded]
            return {"calls": calls, "keep": keep, "x": x_slot, "e": e_slot, "st": st_slot, "n_out": n_out, "lens": lens_dev}
        ws_floats = self.lib.raw("ds_fc_workspace_floats")(B, k, n_out)
        if ws_floats <= 0:
            raise RuntimeError(f"ds_fc_workspace_floats({B},{k},{n_out}) failed: {ws_floats}")
        ws, f = buf(ws_floats), buf(B, n_out)
        pooled = buf(B, k)
        if masked:
            calls.append((self.lib.raw("ds_avgpool_time_masked_f32"),
                          (self._p(a), self._p(lens_dev[len(pw.stages) - 1]), self._p(pooled), B, h, w, cin, st_slot), None, 0.0))
        else:
            calls.append((self.lib.raw("ds_avgpool_time_f32"), (self._p(a), self._p(pooled), B, h, w, cin, st_slot), None, 0.0))
        calls.append((self.lib.raw("ds_fc_l2norm_fwd_f32"),
                      (self._p(pooled), self._p(pw.fc), self._p(pw.fc_bias.detach()), self._p(ws), self._p(f), e_slot, B,
                       k, n_out, ALPHA, L2_EPS, st_slot), None, 0.0))
        keep += [pw, folded]
        return {"calls": calls, "keep": keep, "x": x_slot, "e": e_slot, "st": st_slot, "n_out": n_out, "lens": lens_dev}

    def drop_eval_plans(self) -> int:
        """Forget every cached eval launch plan (each pins its activation buffers, its packed filters and its folded
        BatchNorm): for a process that is done with the models / shapes it has been running and is about to run others
        -- the LRU bounds (PLAN_CACHE_ENTRIES / PLAN_CACHE_BYTES) would get there too, one eviction per new plan.
        Returns the number of bytes of activation buffers released (to torch's caching allocator)."""
        plans = self.__dict__.get("_eval_plans", {})
        freed = sum(q.get("bytes", 0) for q in plans.values())
        plans.clear()
        return freed

    def forward_eval_planned(self, x: torch.Tensor, pw: PackedWeights, folded, precision: str = "f32",
                             lengths: Optional[torch.Tensor] = None, low_latency: bool = False) -> torch.Tensor:
        """forward_eval through a launch plan cached per (shape, weights version, precision, device).

        `lengths` (int tensor [B], on the host): x is a zero-padded batch of utterances of these lengths (frames);
        every kept row of every layer -- and hence the embedding -- is then bit-identical to the forward of the
        utterance alone (rows past an utterance's extent are re-zeroed after each layer, the temporal mean runs
        over its own rows).

        `low_latency` (fp16 path): launches too small to fill the GPU split their contraction over several
        workgroups per tile (serving one or a few utterances); results then differ from the one-pass path by the f32
        summation order."""
        self._check(x, "input")
        if x.dim() != 4 or x.shape[1] != 1:
            raise ValueError("input must be [B,1,T,F] (reference model.py:185, SURVEY F1)")
        if precision not in PRECISIONS:
            raise ValueError(f"unknown precision {precision!r}; expected one of {PRECISIONS}")
        if precision == "f16":
            if pw.stages[0].l_conv1_f16 is None:
                raise ValueError("pack_weights(..., with_f16=True) is required for precision 'f16'")
        elif precision != "f32" and pw.stages[0].l_conv1_bf16 is None:
            raise ValueError("pack_weights(..., with_bf16=True) is required for the bf16 precisions")
        # the plan owns its intermediate activations: forwards in flight on different streams need their own
        stream_id = torch.cuda.current_stream(x.device).cuda_stream if x.is_cuda else 0
        key = (tuple(x.shape), precision, id(pw), id(folded), x.device, stream_id, lengths is not None, low_latency)
        plans = self.__dict__.setdefault("_eval_plans", {})
        plan = plans.pop(key, None)
        if plan is not None:
            plans[key] = plan                           # most recently used last
        if plan is None:
            # a plan pins its packed filters, folded BatchNorm and activation buffers: drop EVERY plan (any shape) that
            # was built from an older generation of these same weights -- i.e. for the same precision and device but
            # another packed-filter / folded-BatchNorm object of a model whose current objects are `pw` / `folded` --
            # and bound the total.  (Plans of other live models keep their own generation: owners are tracked.)
            owner = getattr(pw, "owner", None)
            for k in [k for k, q in plans.items()
                      if k[1] == key[1] and k[4] == key[4] and k[2:4] != key[2:4]
                      and (q.get("owner") is None or q.get("owner") == owner)]:
                del plans[k]
            plan = self._build_eval_plan(x, pw, folded, precision, masked=lengths is not None,
                                         low_latency=low_latency and precision == "f16")
            plan["bytes"] = sum(t.numel() * t.element_size() for t in plan["keep"] if isinstance(t, torch.Tensor))
            plan["owner"] = owner
            # least recently used first (dicts keep insertion order): at most PLAN_CACHE_ENTRIES plans /
            # PLAN_CACHE_BYTES of activation buffers (class attributes; lower them on a GPU shared with other work)
            while plans and (len(plans) >= self.PLAN_CACHE_ENTRIES
                             or sum(q["bytes"] for q in plans.values()) + plan["bytes"] > self.PLAN_CACHE_BYTES):
                del plans[next(iter(plans))]
            plans[key] = plan
        if lengths is not None:
            ln = lengths.to(torch.int64).cpu()
            if ln.numel() != x.shape[0] or int(ln.min()) < 1 or int(ln.max()) > x.shape[2]:
                raise ValueError("lengths must hold one frame count in [1, T] per utterance of the padded batch")
            per_stage = []
            for _ in range(len(pw.stages)):
                ln = (ln - 1) // 2 + 1                  # rows after a 5x5 stride-2 pad-2 layer
                per_stage.append(ln)
            host = torch.stack(per_stage).to(torch.int32)
            if x.is_cuda:
                host = host.pin_memory()
            plan["lens"].copy_(host, non_blocking=True)
            plan["lens_host"] = host                    # keeps the pinned staging copy alive until the next call
        e = torch.empty((x.shape[0], plan["n_out"]), dtype=torch.float32, device=x.device)
        plan["x"].value, plan["e"].value = x.data_ptr(), e.data_ptr()
        plan["st"].value = torch.cuda.current_stream(x.device).cuda_stream if x.is_cuda else None
        prof = self.profile if (self.profile is not None and x.is_cuda) else None
        if prof is not None and self.profile_every > 1:
            turn = self._profile_calls.get(precision, 0)
            self._profile_calls[precision] = turn + 1
            if turn % self.profile_every:
                prof = None
        roctx = _Range.enabled and x.is_cuda
        if roctx:
            torch.cuda.nvtx.range_push(f"ds.forward_eval[{precision}] B={x.shape[0]} T={x.shape[2]}")
        for fn, args, label, flops in plan["calls"]:
            if roctx:
                torch.cuda.nvtx.mark(label or fn.__name__)
            if prof is not None and label is not None:
                if self.self_timed_launches and fn.__name__ in _SELF_TIMED:    # one MFMA kernel per call: the launch carries its own events
                    ev0, ev1 = LaunchEvent(self.lib), LaunchEvent(self.lib)
                    self.lib.call("ds_launch_timing_arm", ev0.handle, ev1.handle)
                    rc = fn(*args)
                    if self.lib.raw("ds_launch_timing_end")() != 1 and rc == 0:
                        raise RuntimeError(f"{fn.__name__}: expected exactly one timed launch")
                else:
                    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    ev0.record()
                    rc = fn(*args)
                    ev1.record()
                prof.append((label, flops, ev0, ev1, precision))
            else:
                rc = fn(*args)
            if rc != 0:
                if roctx:
                    torch.cuda.nvtx.range_pop()
                raise RuntimeError(f"{fn.__name__} failed: {rc} ({self.lib.error_string(rc)})")
        if roctx:
            torch.cuda.nvtx.range_pop()
        return e

    # ------------------------------------------------------------------ forward passes
    def forward_eval(self, x: torch.Tensor, pw: PackedWeights, folded: Dict[str, Tuple[torch.Tensor, torch.Tensor]],
                     taps: Optional[dict] = None, precision: str = "f32") -> torch.Tensor:
        """Eval-mode forward: BatchNorm (running statistics), residual add and clipped ReLU all live in
        the convolution epilogues -- 3 launches per stage, no intermediate normalisation pass.

        precision: "f32" exact-f32 MFMA (the parity path); "bf16x3" split-operand bf16 MFMA (f32-class
        accuracy); "f16" fp16 operands and fp16 activations in HBM, f32 accumulate (the throughput path, 3.7e-4
        from the reference); "bf16" plain bf16 operands (3e-3: outside the contract).  conv1 runs its
        split-operand bf16 kernel in all low-precision modes; fc and the tail are f32 in all modes."""
        if precision not in PRECISIONS:
            raise ValueError(f"unknown precision {precision!r}; expected one of {PRECISIONS}")
        lowp = precision != "f32"
        x3 = precision == "bf16x3"
        h16 = precision == "f16"
        if h16:
            if pw.stages[0].l_conv1_f16 is None:
                raise ValueError("pack_weights(..., with_f16=True) is required for precision 'f16'")
        elif lowp and pw.stages[0].l_conv1_bf16 is None:
            raise ValueError("pack_weights(..., with_bf16=True) is required for the bf16 precisions")
        self._check(x, "input")
        B, one, T, F = x.shape
        if one != 1:
            raise ValueError("input must be [B,1,T,F] (reference model.py:185, SURVEY F1)")
        h, w, cin = T, F, 1
        a = x
        AC = DS_EPI_AFFINE | DS_EPI_CLIP
        for s, sw in enumerate(pw.stages):
            i, c = s + 1, STAGE_CHANNELS[s]
            sc, sh = folded[f"model.bn{i}"]
            if i == 1:
                a, _ = self.conv1(a, sw.conv, B, h, w, sc, sh, AC | (DS_EPI_OUT_F16 if h16 else 0), lowp=lowp)
            elif h16:
                a = self.conv_f16(a, sw.conv_f16, B, h, w, cin, c, 5, 2, sc, sh, None, AC)
            elif lowp:
                a = self.conv_bf16(a, sw.conv_bf16, x3, B, h, w, cin, c, 5, 2, sc, sh, None, AC)
            else:
                a, _ = self.conv(a, sw.conv, B, h, w, cin, c, 5, 2, sc, sh, None, AC)
            h, w, cin = conv_out(h, 5, 2), conv_out(w, 5, 2), c
            if taps is not None:
                taps[f"stage{i}.a"] = a
            sc, sh = folded[f"model.layer{i}.0.bn1"]
            if h16:
                y = self.conv_f16(a, sw.l_conv1_f16, B, h, w, c, c, 3, 1, sc, sh, None, AC)
            elif lowp:
                y = self.conv_bf16(a, sw.l_conv1_bf16, x3, B, h, w, c, c, 3, 1, sc, sh, None, AC)
            else:
                y, _ = self.conv(a, sw.l_conv1, B, h, w, c, c, 3, 1, sc, sh, None, AC)
            if taps is not None:
                taps[f"stage{i}.b"] = y
            sc, sh = folded[f"model.layer{i}.0.bn2"]
            if h16:         # the last layer hands f32 to the (f32) pooling / projection tail
                last = DS_EPI_OUT_F32 if s == len(pw.stages) - 1 else 0
                a = self.conv_f16(y, sw.l_conv2_f16, B, h, w, c, c, 3, 1, sc, sh, a, AC | DS_EPI_RESIDUAL | last)
            elif lowp:
                a = self.conv_bf16(y, sw.l_conv2_bf16, x3, B, h, w, c, c, 3, 1, sc, sh, a, AC | DS_EPI_RESIDUAL)
            else:
                a, _ = self.conv(y, sw.l_conv2, B, h, w, c, c, 3, 1, sc, sh, a, AC | DS_EPI_RESIDUAL)
            if taps is not None:
                taps[f"stage{i}.c"] = a
        return self.tail(a, pw)

    def forward_train(self, x: torch.Tensor, pw: PackedWeights, bns: Dict[str, BNParams],
                      save: bool = True, reducer=None, precision: str = "f32") -> Tuple[torch.Tensor, Optional[SavedForward]]:
        """Train-mode forward: each convolution emits its raw output plus per-tile column sums, a
        finalize kernel turns them into batch statistics (and updates the running ones), and one
        elementwise pass normalises + adds the residual + clips (nn.BatchNorm2d.train() semantics)."""
        self._check(x, "input")
        B, one, T, F = x.shape
        if one != 1:
            raise ValueError("input must be [B,1,T,F]")
        saved = SavedForward(x=x) if save else None
        if precision not in ("f32", "bf16x3"):
            raise ValueError("training runs in f32 or bf16x3 (the plain-bf16 speed mode is eval-only)")
        x3 = precision == "bf16x3"
        if x3 and pw.stages[0].l_conv1_bf16 is None:
            raise ValueError("pack_weights(..., with_bf16=True) is required for bf16x3")

        def conv_s(src, w_f32, w_b, Bc, hh, ww, ci, co, ks, stride):
            if x3:
                return self.conv_bf16(src, w_b, True, Bc, hh, ww, ci, co, ks, stride, want_stats=True)
            return self.conv(src, w_f32, Bc, hh, ww, ci, co, ks, stride, want_stats=True)

        h, w, cin = T, F, 1
        a = x
        for s, sw in enumerate(pw.stages):
            i, c = s + 1, STAGE_CHANNELS[s]
            if i == 1:
                z, st = self.conv1(a, sw.conv, B, h, w, want_stats=True, lowp=x3)
            else:
                z, st = conv_s(a, sw.conv, sw.conv_bf16, B, h, w, cin, c, 5, 2)
            h, w, cin = conv_out(h, 5, 2), conv_out(w, 5, 2), c
            count = B * h * w
            name = f"model.bn{i}"
            mean, invstd, sc, sh = self.bn_finalize(st, count, bns[name], reducer=reducer)
            a = self.bn_apply(z, sc, sh, None, DS_EPI_CLIP)
            if save:
                saved.raws[name], saved.stats[name], saved.acts[f"stage{i}.a"] = z, (mean, invstd, sc, sh), a
            name = f"model.layer{i}.0.bn1"
            z, st = conv_s(a, sw.l_conv1, sw.l_conv1_bf16, B, h, w, c, c, 3, 1)
            mean, invstd, sc, sh = self.bn_finalize(st, count, bns[name], reducer=reducer)
            y = self.bn_apply(z, sc, sh, None, DS_EPI_CLIP)
            if save:
                saved.raws[name], saved.stats[name], saved.acts[f"stage{i}.b"] = z, (mean, invstd, sc, sh), y
            name = f"model.layer{i}.0.bn2"
            z, st = conv_s(y, sw.l_conv2, sw.l_conv2_bf16, B, h, w, c, c, 3, 1)
            mean, invstd, sc, sh = self.bn_finalize(st, count, bns[name], reducer=reducer)
            a = self.bn_apply(z, sc, sh, a, DS_EPI_CLIP | DS_EPI_RESIDUAL)
            if save:
                saved.raws[name], saved.stats[name], saved.acts[f"stage{i}.c"] = z, (mean, invstd, sc, sh), a
                saved.dims.append((h, w))
        e = self.tail(a, pw, saved)
        return e, saved

    # ------------------------------------------------------------------ grouped train-mode forward
    def _group_aligned(self, shp: ConvShape, members: int, x3: bool) -> bool:
        """True if one launch over the concatenated batch keeps every M tile inside one member, i.e. the
        per-tile BatchNorm partial sums can be split by member."""
        out8 = (ctypes.c_int * 8)()
        if x3:
            rc = self.lib.raw("ds_conv_bf16_plan_describe")(ctypes.byref(shp), 1, out8)
        else:
            rc = self.lib.raw("ds_conv_plan_describe")(ctypes.byref(shp), out8)
        if rc != 0:
            return False
        rt, ni = out8[2], out8[3]
        ho = conv_out(shp.H, shp.KS, shp.stride)
        segs_per_member = (shp.B // members) * ((ho + rt - 1) // rt)
        return segs_per_member % ni == 0

    def forward_train_group(self, xs: List[torch.Tensor], pw: PackedWeights, bns: Dict[str, BNParams],
                            save: bool = True, reducer=None, precision: str = "f32"):
        """The reference's three train-mode forwards of one step -- model(data_a), model(data_p), model(data_n),
        train_triplet.py:215 -- run in lock-step over ONE concatenated batch: same arithmetic per utterance, one
        BatchNorm statistic set (and one running-statistics update, in call order) per member (equal to three
        calls up to the order in which per-tile partial sums are folded), but

          * each layer is one convolution launch over all members wherever its tiles do not straddle members
            (otherwise one launch per member, into slices of the same buffer),
          * the backward pass sees one batch: ONE data-gradient and ONE filter-gradient launch per layer, whose
            contraction over all members' pixels IS the sum the reference gets by accumulating three backward
            passes into `.grad`,
          * data-parallel training exchanges the statistics of all members in one all-reduce per BatchNorm
            layer (24 per step instead of 72).

        Returns ([embeddings per member], SavedForward of the concatenated batch; `stats[name]` is a list with one
        (mean, invstd, scale, shift) per member)."""
        if precision not in ("f32", "bf16x3"):
            raise ValueError("training runs in f32 or bf16x3")
        G = len(xs)
        for x in xs:
            self._check(x, "input")
            if x.shape != xs[0].shape or x.dim() != 4 or x.shape[1] != 1:
                raise ValueError("members must be equally shaped [B,1,T,F] batches")
        Bm, _, T, F = xs[0].shape
        B = G * Bm
        # members that already are consecutive slices of one buffer (a resident batch split into a / p / n) are used
        # in place; otherwise they are concatenated
        nbytes = xs[0].numel() * xs[0].element_size()
        if all(x.is_contiguous() and x.data_ptr() == xs[0].data_ptr() + g * nbytes
               and x.untyped_storage().data_ptr() == xs[0].untyped_storage().data_ptr() for g, x in enumerate(xs)):
            x = torch.as_strided(xs[0], (B, 1, T, F), xs[0].stride())
        else:
            x = torch.cat(xs)
        x3 = precision == "bf16x3"
        if x3 and pw.stages[0].l_conv1_bf16 is None:
            raise ValueError("pack_weights(..., with_bf16=True) is required for bf16x3")
        # (save=False frees a layer's buffers while other members' streams may still read them: lock-step then)
        dp = reducer is not None and reducer.active
        if self.MEMBER_STREAMS and save and x.is_cuda and G > 1 and (not dp or self.MEMBER_STREAMS_DP):
            return self._forward_train_group_streams(x, G, pw, bns, save, x3, reducer if dp else None)
        saved = SavedForward(x=x) if save else None
        dev = x.device

        def member(t, g):
            return t[g * Bm:(g + 1) * Bm]

        def conv_g(src, w_f32, w_b, hh, ww, ci, co, ks, stride):
            """raw conv output [B,ho,wo,co] + per-member partial statistics"""
            shp = ConvShape(B, hh, ww, ci, co, ks, stride)
            if G == 1 or self._group_aligned(shp, G, x3):
                if x3:
                    z, st = self.conv_bf16(src, w_b, True, B, hh, ww, ci, co, ks, stride, want_stats=True)
                else:
                    z, st = self.conv(src, w_f32, B, hh, ww, ci, co, ks, stride, want_stats=True)
                rows = st.shape[0] // G
                return z, [st[g * rows:(g + 1) * rows] for g in range(G)]
            ho, wo = conv_out(hh, ks, stride), conv_out(ww, ks, stride)
            z = torch.empty((B, ho, wo, co), dtype=torch.float32, device=dev)
            sts = []
            for g in range(G):
                if x3:
                    zg, st = self.conv_bf16(member(src, g), w_b, True, Bm, hh, ww, ci, co, ks, stride, want_stats=True)
                else:
                    zg, st = self.conv(member(src, g), w_f32, Bm, hh, ww, ci, co, ks, stride, want_stats=True)
                member(z, g).copy_(zg)
                sts.append(st)
            return z, sts

        def bn_g(z, sts, name, count, residual, flags):
            """per-member statistics -> normalised (+ residual) + clipped activations of the whole batch"""
            bn = bns[name]
            c = bn.weight.numel()
            per = []
            if reducer is not None and reducer.active:
                sums = torch.empty((G, 2 * c + 1), dtype=torch.float64, device=dev)
                rows = sts[0].shape[0]
                if all(st_.shape[0] == rows and st_.data_ptr() == sts[0].data_ptr() + g * rows * c * 8
                       for g, st_ in enumerate(sts)):
                    # the members' partial rows are consecutive slices of one launch's statistics: one kernel
                    self.lib.call("ds_partial_sum_f64_group", self._p(sts[0]), rows, self._p(sums), count, c, G,
                                  self._stream(z))
                else:
                    for g in range(G):
                        self.lib.call("ds_partial_sum_f64", self._p(sts[g]), sts[g].shape[0], self._p(sums[g]), c,
                                      self._stream(z))
                    sums[:, 2 * c] = float(count)
                reducer.all_reduce_sum_(sums)                     # all members of this layer in ONE collective
                # mean / invstd / scale / shift as rows of one [G][C] table each (as below): the backward pass stays grouped
                mean_all, invstd_all, sc_all, sh_all = torch.empty((4, G, c), dtype=torch.float32, device=dev).unbind(0)
                for g in range(G):                                # running statistics update in call order
                    mean, invstd, sc, sh = mean_all[g], invstd_all[g], sc_all[g], sh_all[g]
                    self.lib.call("ds_bn_stats_from_sums_f32", self._p(sums[g]), 0, self._p(bn.weight.detach()),
                                  self._p(bn.bias.detach()), BN_EPS, BN_MOMENTUM, self._p(bn.running_mean),
                                  self._p(bn.running_var), self._p(mean), self._p(invstd), self._p(sc), self._p(sh), c,
                                  self._stream(z))
                    per.append((mean, invstd, sc, sh))
            else:
                # the members' mean / invstd (and scale / shift) as rows of one [G][C] tensor each: the backward pass then
                # runs every BatchNorm layer's reductions for all members in one launch (backward._bn_bwd_group), or
                # inside the data-gradient kernel above it (backward._dgrad_bn_bwd)
                mean_all, invstd_all, sc_all, sh_all = torch.empty((4, G, c), dtype=torch.float32, device=dev).unbind(0)
                for g in range(G):                                # running statistics update in call order
                    per.append(self.bn_finalize(sts[g], count, bn, out=(mean_all[g], invstd_all[g], sc_all[g], sh_all[g])))
            a = torch.empty_like(z)
            n_pix = member(z, 0).numel() // c
            for g in range(G):
                self.lib.call("ds_bn_apply_f32", self._p(member(z, g)), self._p(per[g][2]), self._p(per[g][3]),
                              self._p(member(residual, g)) if residual is not None else None, self._p(member(a, g)),
                              n_pix, c, flags, self._stream(z))
            return a, list(per)

        h, w, cin = T, F, 1
        a = x
        for s, sw in enumerate(pw.stages):
            i, c = s + 1, STAGE_CHANNELS[s]
            if i == 1:
                z, st = self.conv1(a, sw.conv, B, h, w, want_stats=True, lowp=x3)     # its tiles never span images
                rows = st.shape[0] // G
                sts = [st[g * rows:(g + 1) * rows] for g in range(G)]
            else:
                z, sts = conv_g(a, sw.conv, sw.conv_bf16, h, w, cin, c, 5, 2)
            h, w, cin = conv_out(h, 5, 2), conv_out(w, 5, 2), c
            count = Bm * h * w
            name = f"model.bn{i}"
            a, stt = bn_g(z, sts, name, count, None, DS_EPI_CLIP)
            if save:
                saved.raws[name], saved.stats[name], saved.acts[f"stage{i}.a"] = z, stt, a
            name = f"model.layer{i}.0.bn1"
            z, sts = conv_g(a, sw.l_conv1, sw.l_conv1_bf16, h, w, c, c, 3, 1)
            y, stt = bn_g(z, sts, name, count, None, DS_EPI_CLIP)
            if save:
                saved.raws[name], saved.stats[name], saved.acts[f"stage{i}.b"] = z, stt, y
            name = f"model.layer{i}.0.bn2"
            z, sts = conv_g(y, sw.l_conv2, sw.l_conv2_bf16, h, w, c, c, 3, 1)
            a, stt = bn_g(z, sts, name, count, a, DS_EPI_CLIP | DS_EPI_RESIDUAL)
            if save:
                saved.raws[name], saved.stats[name], saved.acts[f"stage{i}.c"] = z, stt, a
                saved.dims.append((h, w))
        e = self.tail(a, pw, saved)
        return [member(e, g) for g in range(G)], saved

    MEMBER_STREAMS = True      # forward_train_group on a GPU, single process: one HIP stream per member (see below)
    _member_streams: Dict[Tuple[torch.device, int], list] = {}

    MEMBER_STREAMS_DP = True   # ... also under data parallelism (the members' streams meet at each layer's all-reduce)

    def _forward_train_group_streams(self, x, G: int, pw: PackedWeights, bns: Dict[str, BNParams], save: bool, x3: bool,
                                     reducer=None):
        """forward_train_group with every member's chain -- convolution, statistics, normalise + clip, layer after
        layer -- on a HIP stream of its own.  A train-mode layer is a matrix-core-bound convolution followed by
        HBM-bound element-wise passes that need the whole member's statistics first; in lock-step over one batch the
        chip alternates between the two.  Members are independent until the loss, so on separate streams one
        member's BatchNorm passes run next to another member's convolution (measured on the 768-utterance step:
        6.9 -> 6.2 ms per forward, tools/train_fwd_streams.py).  Same kernels on the same slices of the same buffers:
        the saved state is the lock-step one (one activation buffer per layer, statistics as [G][C] tables), the
        backward pass does not change.  The running statistics are updated in call order (a, p, n): a member's
        finalize kernel waits for the previous member's of the same layer.  Everything that outlives the forward is
        allocated on the caller's stream before the fork; the side streams only allocate what they consume
        themselves.  Data-parallel training (`reducer`): the members' streams meet at each layer's all-reduce, which
        still carries all three members' sums in one collective (member 0's stream issues it after the other two have
        folded their partial sums), and part again for their normalise + clip passes -- 20.55 -> 19.76 ms per step with
        every collective forced on one rank (19.2 without collectives, same box).  (Measured and dropped: an all-reduce
        per member from its own stream, 23.0 ms -- a process group runs its collectives on one internal stream in issue
        order; HIP stream priorities to stagger the members, 29 ms.)"""
        B, _, T, F = x.shape
        Bm = B // G
        dev = x.device
        cur = torch.cuda.current_stream(dev)
        key = (dev, G)
        if key not in Engine._member_streams:
            Engine._member_streams[key] = [torch.cuda.Stream(device=dev) for _ in range(G)]
        streams = Engine._member_streams[key]
        saved = SavedForward(x=x) if save else None
        keep = []                  # data-parallel: the layers' sum tables, alive until the streams have joined
        for st in streams:
            st.wait_stream(cur)

        def member(t, g):
            return t[g * Bm:(g + 1) * Bm]

        def layer(src, name, kind, sw_f32, sw_b, hh, ww, ci, co, residual):
            """one convolution + BatchNorm(train) + clip layer of all members; returns (z, a, stats list)"""
            ks, stride = (5, 2) if kind != "3x3" else (3, 1)
            ho, wo = conv_out(hh, ks, stride), conv_out(ww, ks, stride)
            z = torch.empty((B, ho, wo, co), dtype=torch.float32, device=dev)
            a = torch.empty_like(z)
            tables = torch.empty((4, G, co), dtype=torch.float32, device=dev)
            mean_all, invstd_all, sc_all, sh_all = tables.unbind(0)
            bn = bns[name]
            count = Bm * ho * wo
            flags = DS_EPI_CLIP | (DS_EPI_RESIDUAL if residual is not None else 0)
            prev_done = None
            per = []
            if reducer is not None:
                # data parallel: ONE all-reduce carries the three members' sums of this layer -- the members' streams
                # meet there (member 0's stream issues it) and part again for their normalise + clip passes
                sums = torch.empty((G, 2 * co + 1), dtype=torch.float64, device=dev)
                keep.append(sums)
                ready = []
                for g in range(G):
                    with torch.cuda.stream(streams[g]):
                        if kind == "c1":
                            _, stp = self.conv1(member(src, g), sw_f32, Bm, hh, ww, want_stats=True, lowp=x3, out=member(z, g))
                        elif x3:
                            _, stp = self.conv_bf16(member(src, g), sw_b, True, Bm, hh, ww, ci, co, ks, stride,
                                                    want_stats=True, out=member(z, g))
                        else:
                            _, stp = self.conv(member(src, g), sw_f32, Bm, hh, ww, ci, co, ks, stride, want_stats=True,
                                               out=member(z, g))
                        self.lib.call("ds_partial_sum_f64_group", self._p(stp), stp.shape[0], self._p(sums[g]), count, co,
                                      1, self._stream(z))
                        ev = torch.cuda.Event()
                        ev.record(streams[g])
                        ready.append(ev)
                with torch.cuda.stream(streams[0]):
                    for ev in ready[1:]:
                        streams[0].wait_event(ev)
                    reducer.all_reduce_sum_(sums)
                    reduced = torch.cuda.Event()
                    reduced.record(streams[0])
                for g in range(G):
                    with torch.cuda.stream(streams[g]):
                        streams[g].wait_event(reduced)
                        if prev_done is not None:
                            streams[g].wait_event(prev_done)      # running statistics: a, then p, then n
                        self.lib.call("ds_bn_stats_from_sums_f32", self._p(sums[g]), 0, self._p(bn.weight.detach()),
                                      self._p(bn.bias.detach()), BN_EPS, BN_MOMENTUM, self._p(bn.running_mean),
                                      self._p(bn.running_var), self._p(mean_all[g]), self._p(invstd_all[g]),
                                      self._p(sc_all[g]), self._p(sh_all[g]), co, self._stream(z))
                        prev_done = torch.cuda.Event()
                        prev_done.record(streams[g])
                        self.bn_apply(member(z, g), sc_all[g], sh_all[g],
                                      member(residual, g) if residual is not None else None, flags, out=member(a, g))
                        per.append((mean_all[g], invstd_all[g], sc_all[g], sh_all[g]))
                return z, a, per, ho, wo
            for g in range(G):
                with torch.cuda.stream(streams[g]):
                    if kind == "c1":
                        _, stp = self.conv1(member(src, g), sw_f32, Bm, hh, ww, want_stats=True, lowp=x3, out=member(z, g))
                    elif x3:
                        _, stp = self.conv_bf16(member(src, g), sw_b, True, Bm, hh, ww, ci, co, ks, stride,
                                                want_stats=True, out=member(z, g))
                    else:
                        _, stp = self.conv(member(src, g), sw_f32, Bm, hh, ww, ci, co, ks, stride, want_stats=True,
                                           out=member(z, g))
                    if prev_done is not None:
                        streams[g].wait_event(prev_done)          # running statistics: a, then p, then n
                    out = (mean_all[g], invstd_all[g], sc_all[g], sh_all[g])
                    self.bn_finalize(stp, count, bn, out=out)
                    prev_done = torch.cuda.Event()
                    prev_done.record(streams[g])
                    self.bn_apply(member(z, g), sc_all[g], sh_all[g], member(residual, g) if residual is not None else None,
                                  flags, out=member(a, g))
                    per.append(out)
            return z, a, per, ho, wo

        h, w, cin = T, F, 1
        a = x
        for s_, sw in enumerate(pw.stages):
            i, c = s_ + 1, STAGE_CHANNELS[s_]
            name = f"model.bn{i}"
            z, a, stt, h, w = layer(a, name, "c1" if i == 1 else "5x5", sw.conv, sw.conv_bf16, h, w, cin, c, None)
            cin = c
            if save:
                saved.raws[name], saved.stats[name], saved.acts[f"stage{i}.a"] = z, stt, a
            name = f"model.layer{i}.0.bn1"
            z, y, stt, _, _ = layer(a, name, "3x3", sw.l_conv1, sw.l_conv1_bf16, h, w, c, c, None)
            if save:
                saved.raws[name], saved.stats[name], saved.acts[f"stage{i}.b"] = z, stt, y
            name = f"model.layer{i}.0.bn2"
            z, a, stt, _, _ = layer(y, name, "3x3", sw.l_conv2, sw.l_conv2_bf16, h, w, c, c, a)
            if save:
                saved.raws[name], saved.stats[name], saved.acts[f"stage{i}.c"] = z, stt, a
                saved.dims.append((h, w))
        for st in streams:
            cur.wait_stream(st)
        e = self.tail(a, pw, saved)
        return [member(e, g) for g in range(G)], saved

    # ------------------------------------------------------------------ loss side
    def pairwise_distance(self, x1: torch.Tensor, x2: torch.Tensor) -> torch.Tensor:
        assert x1.size() == x2.size()                       # reference model.py:14
        self._check(x1, "x1"), self._check(x2, "x2")
        n, d = x1.shape
        out = torch.empty(n, dtype=torch.float32, device=x1.device)
        self.lib.call("ds_pairwise_distance_f32", self._p(x1), self._p(x2), self._p(out), n, d, self._stream(x1))
        return out

    def triplet_tail(self, a, p, n, margin: float, band: float = 0.0, amb_cap: int = 0, probe_base: int = -1) -> dict:
        """The loss side of one triplet step (model.py:27-33, train_triplet.py:251-262) in two launches:
        distances, then one scan giving the loss, the ordered filter, mean(d_n - d_p) and (amb_cap > 0) the
        near-tie list (probe_base >= 0: the slots near ties leave unused hold the probe triplets probe_base, probe_base + 1,
        ... mod N, see mining.RefinePolicy).  Nothing is cached between calls: buffers that are rewritten through raw pointers (a HIP
        graph's static output, a collective's destination) keep their address AND their torch version counter, so a
        result keyed on those would be served for new contents."""
        for t, nm in ((a, "anchor"), (p, "positive"), (n, "negative")):
            self._check(t, nm)
        assert a.size() == p.size() == n.size()
        rows, d = a.shape
        dev = a.device
        out = {"d_p": torch.empty(rows, dtype=torch.float32, device=dev),
               "d_n": torch.empty(rows, dtype=torch.float32, device=dev),
               "loss": torch.empty(1, dtype=torch.float32, device=dev),
               "idx": torch.empty(rows, dtype=torch.int64, device=dev),
               "count": torch.empty(1, dtype=torch.int32, device=dev),
               "mean_diff": torch.empty(1, dtype=torch.float32, device=dev),
               "amb_idx": torch.empty(amb_cap, dtype=torch.int64, device=dev) if amb_cap > 0 else None,
               "amb_count": torch.empty(1, dtype=torch.int32, device=dev) if amb_cap > 0 else None}
        self.lib.call("ds_triplet_tail_probe_f32", self._p(a), self._p(p), self._p(n), float(margin), float(band),
                      self._p(out["d_p"]), self._p(out["d_n"]), self._p(out["loss"]), self._p(out["idx"]),
                      self._p(out["count"]), self._p(out["mean_diff"]), self._p(out["amb_idx"]),
                      self._p(out["amb_count"]), int(amb_cap), int(probe_base) if amb_cap > 0 else -1, rows, d,
                      self._stream(a))
        return out

    def triplet_margin(self, a, p, n, margin: float):
        t = self.triplet_tail(a, p, n, margin)
        return t["loss"], t["d_p"], t["d_n"]

    def triplet_filter(self, d_p: torch.Tensor, d_n: torch.Tensor, margin: float):
        """train_triplet.py:251-262 on the device: returns (idx[int64, N] of which the first `count`
        entries are valid & ascending, count[int32,1], mean_diff[float32,1]) without synchronising."""
        n = d_p.numel()
        idx = torch.empty(n, dtype=torch.int64, device=d_p.device)
        count = torch.empty(1, dtype=torch.int32, device=d_p.device)
        mean_diff = torch.empty(1, dtype=torch.float32, device=d_p.device)
        self.lib.call("ds_triplet_filter_f32", self._p(d_p), self._p(d_n), float(margin), self._p(idx),
                      self._p(count), self._p(mean_diff), n, self._stream(d_p))
        return idx, count, mean_diff
