"""Kernel logic on the host emulator (no GPU): the UNMODIFIED .hip sources, compiled for the fiber
emulator in tests/emul/, are driven through the same C ABI and compared with the oracle.  These
tests debug indexing, fragment layouts, masks and epilogues; numerical parity on the real
hardware is asserted by the -m gpu tests."""
import ctypes

import numpy as np
import pytest

import deepspeaker_oracle as O
from conftest import rel_err
from emul_util import aligned, emul_lib, nchw, nhwc, ptr, to_aligned
from deepspeaker_pytorch_amd._native import (ConvShape, DS_EPI_AFFINE, DS_EPI_CLIP, DS_EPI_RESIDUAL,
                                             DS_EPI_STATS)


def pack_conv(lib, w, dgrad=0):
    co, ci, k, _ = w.shape
    out, src = aligned(w.size), to_aligned(w)
    lib.call("ds_pack_conv_weight_f32", ptr(src), ptr(out), co, ci, k, dgrad, None)
    return out


def run_conv(lib, x_nchw, w, stride, flags=0, scale=None, shift=None, res_nchw=None, want_stats=False):
    b, ci, h, wd = x_nchw.shape
    co, _, k, _ = w.shape
    shp = ConvShape(b, h, wd, ci, co, k, stride)
    ho, wo = ctypes.c_int(), ctypes.c_int()
    lib.call("ds_conv_out_dims", ctypes.byref(shp), ctypes.byref(ho), ctypes.byref(wo))
    y = aligned((b, ho.value, wo.value, co), fill=np.nan)
    stats = None
    if want_stats:
        flags |= DS_EPI_STATS
        rows = lib.raw("ds_conv_stats_rows")(ctypes.byref(shp))
        assert rows > 0
        stats = aligned((rows, co, 2), fill=np.nan)
    # keep every temporary alive across the call: ptr() only returns an address
    xh, wp = nhwc(x_nchw), pack_conv(lib, w)
    sc = to_aligned(scale) if scale is not None else None
    sh = to_aligned(shift) if shift is not None else None
    rh = nhwc(res_nchw) if res_nchw is not None else None
    lib.call("ds_conv_fwd_f32", ctypes.byref(shp), ptr(xh), ptr(wp), ptr(sc), ptr(sh), ptr(rh), ptr(y),
             ptr(stats), flags, None)
    return nchw(y), stats


CASES = [
    # (B, Cin, Cout, H, W, KS, stride)       what it exercises
    (2, 8, 64, 9, 32, 3, 1),                 # stage-1 geometry, partial last row block (9 = 2*4+1)
    (1, 16, 64, 8, 16, 3, 1),                # two channel chunks, 8-row segments
    (3, 8, 128, 20, 8, 3, 1),                # stage-3 geometry: whole image = 160 rows, 160x128 tile
    (5, 8, 128, 10, 4, 3, 1),                # stage-4 geometry: 4 images per tile, ragged last tile
    (2, 8, 64, 16, 32, 5, 2),                # 5x5 stride 2, even sizes
    (2, 8, 128, 13, 16, 5, 2),               # 5x5 stride 2, odd height (variable-length utterances)
    (3, 16, 128, 7, 8, 5, 2),                # 5x5 s2 into a 4x4 map, multi-image tiles
    (70, 24, 64, 1, 1, 1, 1),                # 1x1 on [B,1,1,C]: the fc GEMM shape class
]


@pytest.mark.parametrize("case", CASES)
def test_conv_fwd_raw(case):
    lib = emul_lib()
    b, ci, co, h, w, k, s = case
    rs = np.random.RandomState(hash(case) % 2**31)
    x = rs.randn(b, ci, h, w).astype(np.float32)
    wt = (rs.randn(co, ci, k, k) / np.sqrt(ci * k * k)).astype(np.float32)
    y, stats = run_conv(lib, x, wt, s, want_stats=True)
    ref = O.conv2d(x.astype(np.float64), wt.astype(np.float64), s, k // 2)
    assert y.shape == ref.shape
    assert rel_err(y, ref) < 2e-6
    # raw statistics: column sums over every real output pixel, nothing else
    tot = stats.astype(np.float64).sum(axis=0)
    assert np.isfinite(tot).all()
    np.testing.assert_allclose(tot[:, 0], ref.sum(axis=(0, 2, 3)), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(tot[:, 1], (ref * ref).sum(axis=(0, 2, 3)), rtol=1e-4, atol=1e-4)


def test_conv_fwd_epilogue():
    lib = emul_lib()
    rs = np.random.RandomState(3)
    b, ci, co, h, w = 2, 8, 64, 6, 16
    x = rs.randn(b, ci, h, w).astype(np.float32) * 3
    wt = rs.randn(co, ci, 3, 3).astype(np.float32)
    scale = rs.uniform(0.5, 1.5, co).astype(np.float32)
    shift = rs.randn(co).astype(np.float32)
    res = rs.randn(b, co, h, w).astype(np.float32) * 5
    y, _ = run_conv(lib, x, wt, 1, DS_EPI_AFFINE | DS_EPI_RESIDUAL | DS_EPI_CLIP, scale, shift, res)
    z = O.conv2d(x.astype(np.float64), wt.astype(np.float64), 1, 1)
    ref = np.clip(z * scale[None, :, None, None] + shift[None, :, None, None] + res, 0, 20)
    assert ((ref == 0).any() and (ref == 20).any())          # both clip edges are hit
    assert np.abs(y - ref).max() < 1e-4


def test_conv_rejects_bad_arguments():
    lib = emul_lib()
    shp = ConvShape(1, 8, 8, 7, 64, 3, 1)                     # Cin not a multiple of 8
    buf = aligned(64 * 64 * 9)
    rc = lib.raw("ds_conv_fwd_f32")(ctypes.byref(shp), ptr(buf), ptr(buf), None, None, None, ptr(buf), None, 0, None)
    assert rc == -1
    shp = ConvShape(1, 8, 8, 8, 64, 3, 1)
    rc = lib.raw("ds_conv_fwd_f32")(ctypes.byref(shp), None, ptr(buf), None, None, None, ptr(buf), None, 0, None)
    assert rc == -3
    rc = lib.raw("ds_conv_fwd_f32")(ctypes.byref(shp), ptr(buf), ptr(buf), None, None, None, ptr(buf), None,
                                    DS_EPI_AFFINE, None)
    assert rc == -3                                           # affine epilogue without scale/shift
    shp = ConvShape(1, 8, 8, 8, 64, 4, 1)
    rc = lib.raw("ds_conv_fwd_f32")(ctypes.byref(shp), ptr(buf), ptr(buf), None, None, None, ptr(buf), None, 0, None)
    assert rc == -4
    assert lib.error_string(-4) == "unsupported configuration"


@pytest.mark.parametrize("shape", [(2, 160, 64), (1, 13, 64), (3, 7, 20)])
def test_conv1(shape):
    lib = emul_lib()
    b, h, w = shape
    rs = np.random.RandomState(h)
    x = rs.randn(b, 1, h, w).astype(np.float32)
    wt = (rs.randn(64, 1, 5, 5) * 0.3).astype(np.float32)
    scale = rs.uniform(0.5, 1.5, 64).astype(np.float32)
    shift = rs.randn(64).astype(np.float32)
    wp, wsrc = aligned(25 * 64), to_aligned(wt)
    lib.call("ds_pack_conv1_weight_f32", ptr(wsrc), ptr(wp), 64, None)
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    y = aligned((b, ho, wo, 64), fill=np.nan)
    rows = lib.raw("ds_conv5x5s2_c1_stats_rows")(b, h)
    stats = aligned((rows, 64, 2), fill=np.nan)
    xa, sc, sh = to_aligned(x), to_aligned(scale), to_aligned(shift)
    lib.call("ds_conv5x5s2_c1_fwd_f32", ptr(xa), ptr(wp), ptr(sc), ptr(sh),
             ptr(y), ptr(stats), b, h, w, 64, DS_EPI_AFFINE | DS_EPI_CLIP | DS_EPI_STATS, None)
    z = O.conv2d(x.astype(np.float64), wt.astype(np.float64), 2, 2)
    ref = np.clip(z * scale[None, :, None, None] + shift[None, :, None, None], 0, 20)
    assert np.abs(nchw(y) - ref).max() < 1e-5
    tot = stats.astype(np.float64).sum(axis=0)
    np.testing.assert_allclose(tot[:, 0], z.sum(axis=(0, 2, 3)), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(tot[:, 1], (z * z).sum(axis=(0, 2, 3)), rtol=1e-4, atol=1e-4)
