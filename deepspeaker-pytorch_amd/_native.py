"""ctypes binding of libdeepspeaker_hip.so (the C ABI declared in include/deepspeaker_hip.h).

The binding is pure plumbing: pointers and sizes in, return codes out.  There is no CPU or
PyTorch fallback -- if the shared library is missing, `load()` raises."""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_longlong, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = "libdeepspeaker_hip.so"

DS_EPI_AFFINE, DS_EPI_RESIDUAL, DS_EPI_CLIP, DS_EPI_STATS, DS_EPI_OUT_F32, DS_EPI_OUT_F16 = 1, 2, 4, 8, 16, 32
DS_EPI_OUT_PLANES16, DS_CONV_IN_PLANES16 = 256, 512
DS_CONV_HINT_SINGLE_BUFFER = 64
DS_CONV_HINT_CHUNK16 = 128
DS_CONV_HINT_NO_PERSIST = 1024
DS_CONV_HINT_NO_WIDE = 2048
DS_CONV_HINT_ONE_QUEUE = 4096
DS_CONV_CK = 8
DS_TAIL_SMALL_MAX_B = 4


class ConvShape(Structure):
    """struct ds_conv_shape"""
    _fields_ = [("B", c_int), ("H", c_int), ("W", c_int), ("Cin", c_int),
                ("Cout", c_int), ("KS", c_int), ("stride", c_int)]


class PackJob(Structure):
    """struct ds_pack_job"""
    _fields_ = [("w_oihw", c_void_p), ("out", c_void_p), ("out2", c_void_p),
                ("Cout", c_int), ("Cin", c_int), ("KS", c_int), ("mode", c_int)]


class DeepSpeakerHipError(RuntimeError):
    pass


_P = c_void_p
_SIGNATURES = {
    # name: (restype, argtypes)
    "ds_version": (c_int, []),
    "ds_mfma_rate_probe": (c_int, [c_int, c_int, _P, _P, _P]),
    "ds_mfma_rate_probe_data": (c_int, [_P, c_longlong, _P, c_longlong, c_int, _P, _P, _P]),
    "ds_event_create": (c_int, [POINTER(c_void_p)]),
    "ds_event_destroy": (c_int, [_P]),
    "ds_event_elapsed_ms": (c_int, [_P, _P, POINTER(c_float)]),
    "ds_launch_timing_arm": (c_int, [_P, _P]),
    "ds_launch_timing_end": (c_int, []),
    "ds_error_string": (c_char_p, [c_int]),
    "ds_sched_workspace_bytes": (ctypes.c_size_t, []),
    "ds_sched_set_workspace": (c_int, [_P, ctypes.c_size_t]),
    "ds_sched_free_slots": (c_longlong, []),
    "ds_nchw_to_nhwc_f32": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "ds_nhwc_to_nchw_f32": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "ds_pack_conv_weight_f32": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "ds_pack_conv1_weight_f32": (c_int, [_P, _P, c_int, _P]),
    "ds_pack_fc_weight_f32": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    "ds_pack_fc_weight_dgrad_f32": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    "ds_bn_fold_f32": (c_int, [_P, _P, _P, _P, c_float, _P, _P, c_int, _P]),
    "ds_bn_stats_finalize_f32": (c_int, [_P, c_int, c_longlong, _P, _P, c_float, c_float, _P, _P, _P, _P,
                                         _P, _P, c_int, _P]),
    "ds_bn_apply_f32": (c_int, [_P, _P, _P, _P, _P, c_longlong, c_int, c_int, _P]),
    "ds_conv_out_dims": (c_int, [POINTER(ConvShape), POINTER(c_int), POINTER(c_int)]),
    "ds_conv_stats_rows": (c_int, [POINTER(ConvShape)]),
    "ds_conv_plan_describe": (c_int, [POINTER(ConvShape), POINTER(c_int)]),
    "ds_conv5x5s2_c1_stats_rows": (c_int, [c_int, c_int]),
    "ds_conv5x5s2_c1_fwd_f32": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "ds_conv5x5s2_c1_fwd_bf16": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "ds_conv_fwd_f32": (c_int, [POINTER(ConvShape), _P, _P, _P, _P, _P, _P, _P, c_int, _P]),
    "ds_pack_conv_weight_bf16": (c_int, [_P, _P, _P, c_int, c_int, c_int, _P]),
    "ds_pack_conv_weight_dgrad_bf16": (c_int, [_P, _P, _P, c_int, c_int, c_int, _P]),
    "ds_pack_conv_weight_dgrad_s2_bf16": (c_int, [_P, _P, _P, c_int, c_int, _P]),
    "ds_conv_dgrad_bf16": (c_int, [POINTER(ConvShape), _P, _P, _P, _P, _P]),
    "ds_conv_bf16_stats_rows": (c_int, [POINTER(ConvShape), c_int]),
    "ds_conv_bf16_plan_describe": (c_int, [POINTER(ConvShape), c_int, POINTER(c_int)]),
    "ds_conv_bf16_set_forced_cfg": (None, [c_int]),
    "ds_conv_fwd_bf16": (c_int, [POINTER(ConvShape), _P, _P, _P, _P, _P, _P, _P, _P, c_int, _P]),
    "ds_pack_conv_weight_f16": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    "ds_pack_conv_weights_f16_batch": (c_int, [POINTER(PackJob), c_int, _P]),
    "ds_pack_conv_weights_bf16_batch": (c_int, [POINTER(PackJob), c_int, _P]),
    "ds_conv_fwd_f16": (c_int, [POINTER(ConvShape), _P, _P, _P, _P, _P, _P, c_int, _P]),
    "ds_conv_f16_splitk_workspace_bytes": (c_longlong, [POINTER(ConvShape)]),
    "ds_conv_fwd_f16_splitk": (c_int, [POINTER(ConvShape), _P, _P, _P, _P, _P, _P, c_int, _P, c_longlong, _P]),
    "ds_conv_block_f16_supported": (c_int, [c_int, c_int, c_int, c_int]),
    "ds_conv_block_f16": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "ds_conv_block_f16_masked": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "ds_conv_f16_plan_describe": (c_int, [POINTER(ConvShape), POINTER(c_int)]),
    "ds_conv_f16_plan_describe_hinted": (c_int, [POINTER(ConvShape), c_int, POINTER(c_int)]),
    "ds_conv_f16_set_layout_padding": (None, [c_int]),
    "ds_conv_f16_set_forced_cfg": (None, [c_int]),
    "ds_conv_f16_plan_lds_layout": (c_int, [POINTER(ConvShape), c_int, POINTER(c_int)]),
    "ds_cast_f32_to_f16": (c_int, [_P, _P, c_longlong, _P]),
    "ds_cast_f16_to_f32": (c_int, [_P, _P, c_longlong, _P]),
    "ds_avgpool_time_f32": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "ds_l2norm_scale_f32": (c_int, [_P, _P, c_int, c_int, c_float, c_float, _P]),
    "ds_max_abs_diff_f32": (c_int, [_P, _P, c_longlong, _P, _P]),
    "ds_fc_workspace_floats": (c_longlong, [c_int, c_int, c_int]),
    "ds_fc_l2norm_fwd_f32": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_float, c_float, _P]),
    "ds_pairwise_distance_f32": (c_int, [_P, _P, _P, c_int, c_int, _P]),
    "ds_pairwise_distance_p_f32": (c_int, [_P, _P, _P, c_int, c_int, c_float, _P]),
    "ds_pairwise_distance_p_bwd_f32": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_float, _P]),
    "ds_triplet_margin_fwd_f32": (c_int, [_P, _P, _P, c_float, _P, _P, _P, c_int, c_int, _P]),
    "ds_triplet_filter_f32": (c_int, [_P, _P, c_float, _P, _P, _P, c_int, _P]),
    "ds_triplet_tail_f32": (c_int, [_P, _P, _P, c_float, c_float, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P]),
    "ds_triplet_scan_f32": (c_int, [_P, _P, c_float, _P, _P, _P, _P, c_int, _P]),
    "ds_refine_distances_f32": (c_int, [_P, _P, _P, c_int, _P, _P, c_int, _P]),
    "ds_triplet_tail_probe_f32": (c_int, [_P, _P, _P, c_float, c_float, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int,
                                          c_int, _P]),
    "ds_refine_distances_probe_f32": (c_int, [_P, _P, _P, c_int, _P, _P, _P, _P, _P, _P, _P, c_int, _P, _P]),
    "ds_refine_distances_fused_f32": (c_int, [_P, _P, _P, c_int, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, _P, _P]),
    "ds_pack_conv_dgrad_s2_f32": (c_int, [_P, _P, c_int, c_int, _P]),
    "ds_conv_dgrad_f32": (c_int, [POINTER(ConvShape), _P, _P, _P, _P]),
    "ds_conv_wgrad_workspace_floats": (c_longlong, [POINTER(ConvShape)]),
    "ds_conv_wgrad_f32": (c_int, [POINTER(ConvShape), _P, _P, _P, _P, c_int, _P]),
    "ds_conv_wgrad_bf16_workspace_floats": (c_longlong, [POINTER(ConvShape)]),
    "ds_conv_wgrad_bf16": (c_int, [POINTER(ConvShape), _P, _P, _P, _P, _P]),
    "ds_pack_fc_weight_rows_f32": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    "ds_tail_small_f32": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, c_float, _P]),
    "ds_pack_conv_weight_dgrad_f16": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "ds_bn_f16_partial_rows": (c_int, [c_longlong, c_int]),
    "ds_bn_stats_group_f16": (c_int, [_P, _P, c_longlong, _P, _P, c_float, c_float, _P, _P, _P, _P, _P, _P, c_int, c_int, _P]),
    "ds_bn_apply_group_f16": (c_int, [_P, _P, _P, _P, _P, c_longlong, c_int, c_int, c_int, _P]),
    "ds_bn_bwd_group_f16": (c_int, [_P, c_int, _P, _P, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_longlong,
                                    c_int, c_int, c_int, c_int, c_float, _P]),
    "ds_bn_stats_partial_f16": (c_int, [_P, _P, c_longlong, c_int, c_int, _P]),
    "ds_bn_stats_from_sums_group_f32": (c_int, [_P, _P, _P, c_float, c_float, _P, _P, _P, _P, _P, _P, c_int, c_int, _P]),
    "ds_bn_bwd_group_reduce_f16": (c_int, [_P, c_int, _P, _P, c_int, _P, _P, _P, _P, _P, _P, _P, c_longlong, c_int, c_int,
                                           c_int, c_int, _P]),
    "ds_bn_bwd_group_apply_f16": (c_int, [_P, _P, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_longlong, c_int, c_int,
                                          c_float, _P]),
    "ds_scale_cast_f32_to_f16": (c_int, [_P, _P, c_longlong, c_float, _P]),
    "ds_conv_wgrad_f16_workspace_floats": (c_longlong, [POINTER(ConvShape)]),
    "ds_conv_wgrad_f16": (c_int, [POINTER(ConvShape), _P, _P, _P, _P, c_float, _P]),
    "ds_conv_wgrad_c1_f16": (c_int, [POINTER(ConvShape), _P, _P, _P, _P, c_float, _P]),
    "ds_bn_bwd_partial_rows": (c_int, [c_longlong, c_int]),
    "ds_bn_bwd_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_longlong, c_int, _P]),
    "ds_bn_bwd_group_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_longlong, c_int, c_int, _P]),
    "ds_colsum_f32": (c_int, [_P, _P, c_int, c_int, _P]),
    "ds_partial_sum_f64": (c_int, [_P, c_int, _P, c_int, _P]),
    "ds_partial_sum_f64_group": (c_int, [_P, c_int, _P, c_longlong, c_int, c_int, _P]),
    "ds_bn_bwd_group_reduce_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, c_longlong, c_int, c_int, _P]),
    "ds_bn_bwd_group_apply_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_longlong, c_int, c_int, _P]),
    "ds_conv_dgrad_bnbwd_bf16_rows": (c_int, [POINTER(ConvShape), c_int]),
    "ds_conv_dgrad_bnbwd_bf16": (c_int, [POINTER(ConvShape), _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, _P, _P, _P]),
    "ds_conv_dgrad_s2_bnbwd_bf16_rows": (c_int, [POINTER(ConvShape), c_int]),
    "ds_conv_dgrad_s2_bnbwd_bf16": (c_int, [POINTER(ConvShape), _P, _P, _P, _P, _P, _P, _P, c_int, _P, _P, _P]),
    "ds_bn_bwd_group_finish_f32": (c_int, [_P, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_longlong, c_int, c_int, _P]),
    "ds_bn_stats_from_sums_f32": (c_int, [_P, c_longlong, _P, _P, c_float, c_float, _P, _P, _P, _P, _P, _P, c_int, _P]),
    "ds_bn_bwd_reduce_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_longlong, c_int, _P]),
    "ds_bn_bwd_apply_f32": (c_int, [_P, c_longlong, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_longlong, c_int, _P]),
    "ds_gather_rows_f32": (c_int, [_P, _P, _P, c_int, c_int, _P]),
    "ds_gather_rows3_f32": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, _P]),
    "ds_scatter_add_rows_f32": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "ds_mine_workspace_floats": (c_longlong, [c_int, c_int]),
    "ds_mine_semihard_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P]),
    "ds_fc_ce_fwd_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "ds_cross_entropy_fwd_f32": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, _P]),
    "ds_cross_entropy_bwd_f32": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "ds_group_mean_f32": (c_int, [_P, _P, c_int, c_int, _P]),
    "ds_mask_rows": (c_int, [_P, _P, c_int, c_int, c_longlong, _P]),
    "ds_avgpool_time_masked_f32": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "ds_segment_mean_f32": (c_int, [_P, _P, _P, c_int, _P]),
    "ds_roc_sweep_f32": (c_int, [_P, _P, c_int, c_float, c_float, c_int, c_int, c_int, _P, _P, _P, _P]),
    "ds_assemble_crops_f32": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, _P]),
    "ds_optim_chunk_elems": (c_int, []),
    "ds_fill_bytes": (c_int, [_P, _P, c_int, _P]),
    "ds_optim_step_inc": (c_int, [_P, _P, _P]),
    "ds_adagrad_step_dev_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, ctypes.c_double, ctypes.c_double, c_float, c_float, _P,
                                        _P, _P]),
    "ds_sgd_step_dev_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_float, c_float, c_float, c_float, _P, _P, _P]),
    "ds_adam_step_dev_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_float, ctypes.c_double, ctypes.c_double, c_float,
                                     c_float, _P, _P, _P]),
    "ds_adagrad_step_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_float, c_float, c_float, _P, _P]),
    "ds_sgd_step_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_float, c_float, c_float, c_float, c_int, _P, _P]),
    "ds_nonfinite_flag_f32": (c_int, [_P, c_longlong, _P, _P]),
    "ds_adam_step_f32": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_float, c_float, c_float, c_float, c_float,
                                 c_float, c_float, _P, _P]),
    "ds_pairwise_distance_bwd_f32": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, _P]),
    "ds_triplet_margin_bwd_f32": (c_int, [_P, _P, _P, _P, _P, c_float, _P, _P, _P, _P, c_int, c_int, _P]),
    "ds_l2norm_scale_bwd_f32": (c_int, [_P, _P, _P, c_int, c_int, c_float, c_float, _P]),
    "ds_avgpool_time_bwd_f32": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P]),
}


def exported_symbols():
    """Every entry point include/deepspeaker_hip.h declares."""
    return sorted(_SIGNATURES)


DS_ERR_NO_WORKSPACE = -5
# entry points that launch persistent kernels (tiles drawn from device-side counters in the caller's scheduler workspace)
_PERSISTENT = ("ds_conv_fwd_f16", "ds_conv_fwd_f16_splitk", "ds_conv_block_f16", "ds_conv_block_f16_masked")


class NativeLib:
    """A loaded copy of the C ABI with argument types declared.

    The library owns no device memory: the persistent kernels' tile-scheduling slots live in zeroed buffers this wrapper
    allocates through torch's allocator (`add_sched_workspace`: one int32 tensor of ds_sched_workspace_bytes() per hand-
    over, kept alive for the life of the process) -- eagerly when a model is moved to a device
    (DeepSpeakerModel._apply), and on demand when a persistent launch reports DS_ERR_NO_WORKSPACE."""

    def __init__(self, path: str, host_memory: bool = False):
        # host_memory: tests only -- the library at `path` is the host emulator (tests/emul), its "device" memory is the host's
        self.host_memory = host_memory
        if not os.path.exists(path):
            raise DeepSpeakerHipError(
                f"{path} not found: build it with `make` (hipcc --offload-arch=gfx950); "
                "this package has no fallback path")
        self.path = path
        self.trace = None                       # set to a dict to count calls per entry point (NativeLib.call only)
        self._dll = ctypes.CDLL(path)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(self._dll, name)        # AttributeError if the ABI is incomplete
            fn.restype = res
            fn.argtypes = args
            setattr(self, "_" + name, fn)
        self._sched_workspaces = []
        for name in _PERSISTENT:
            if name in _SIGNATURES:
                setattr(self, "_" + name, self._with_workspace(name, getattr(self, "_" + name)))

    def _with_workspace(self, name, fn):
        def call(*args):
            rc = fn(*args)
            if rc == DS_ERR_NO_WORKSPACE:       # first persistent launch on this device (or every slot taken by graphs)
                self.add_sched_workspace()
                rc = fn(*args)
            return rc
        call.__name__ = name
        return call

    def add_sched_workspace(self, device=None):
        """Hand the tile scheduler one more zeroed buffer on `device` (default: the current device; the host under the
        test emulator).  Not possible inside a stream capture (an allocation there belongs to the graph's pool and its
        zeroing would be a graph node): move the model to its device, or run one forward, before capturing."""
        import torch
        n = int(self._ds_sched_workspace_bytes())
        if not self.host_memory:
            if not torch.cuda.is_available():
                raise DeepSpeakerHipError("no ROCm device: deepspeaker-pytorch_amd computes only on an MI355X")
            dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
            if torch.cuda.is_current_stream_capturing():
                raise DeepSpeakerHipError("the persistent kernels' scheduler workspace on " + str(dev) + " is missing or used "
                                          "up and cannot be allocated inside a stream capture: call "
                                          "NativeLib.add_sched_workspace() (or run one eager forward) before capturing")
            with torch.cuda.device(dev):
                buf = torch.zeros(n // 4, dtype=torch.int32, device=dev)
                torch.cuda.current_stream(dev).synchronize()        # the zeros must have landed before any launch reads them
                rc = self._ds_sched_set_workspace(ctypes.c_void_p(buf.data_ptr()), n)
        else:
            buf = torch.zeros(n // 4 + 16, dtype=torch.int32)
            off = (-buf.data_ptr() % 64) // 4
            buf = buf[off:off + n // 4]
            rc = self._ds_sched_set_workspace(ctypes.c_void_p(buf.data_ptr()), n)
        if rc != 0:
            raise DeepSpeakerHipError(f"ds_sched_set_workspace failed: {rc} ({self.error_string(rc)})")
        self._sched_workspaces.append(buf)

    def ensure_sched_workspace(self, device, min_free: int = 64):
        """At least `min_free` unassigned slots on `device` (called when a model is moved there: a capture whose first
        persistent launch was never warmed up then finds its slots)."""
        import torch
        dev = torch.device(device)
        if dev.type != "cuda" or self.host_memory:
            return
        with torch.cuda.device(dev):
            if int(self._ds_sched_free_slots()) < min_free:
                self.add_sched_workspace(dev)

    def error_string(self, code: int) -> str:
        return self._ds_error_string(code).decode()

    def call(self, name: str, *args):
        if self.trace is not None:              # tests / tools: which entry points a step really went through
            self.trace[name] = self.trace.get(name, 0) + 1
        rc = getattr(self, "_" + name)(*args)
        if rc != 0:
            raise DeepSpeakerHipError(f"{name} failed: {rc} ({self.error_string(rc)})")
        return rc

    def raw(self, name: str):
        return getattr(self, "_" + name)


_cached = None


def load() -> NativeLib:
    """Load the HIP library that sits next to this file (torch must already be imported so the
    process-wide libamdhip64 is torch's)."""
    global _cached
    if _cached is None:
        _cached = NativeLib(os.path.join(_HERE, LIB_NAME))
    return _cached
