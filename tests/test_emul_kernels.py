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
    # the same layer on the bf16 matrix cores with split operands
    y3 = aligned((b, ho, wo, 64), fill=np.nan)
    stats3 = aligned((rows, 64, 2), fill=np.nan)
    lib.call("ds_conv5x5s2_c1_fwd_bf16", ptr(xa), ptr(wp), ptr(sc), ptr(sh),
             ptr(y3), ptr(stats3), b, h, w, 64, DS_EPI_AFFINE | DS_EPI_CLIP | DS_EPI_STATS, None)
    assert np.abs(nchw(y3) - ref).max() < 3e-4 and rel_err(nchw(y3), ref) < 3e-5
    tot3 = stats3.astype(np.float64).sum(axis=0)
    np.testing.assert_allclose(tot3[:, 0], z.sum(axis=(0, 2, 3)), rtol=1e-3, atol=2e-3)
    np.testing.assert_allclose(tot3[:, 1], (z * z).sum(axis=(0, 2, 3)), rtol=1e-3, atol=2e-3)


# ------------------------------------------------------------------------------------------------
# backward kernels
# ------------------------------------------------------------------------------------------------
DGRAD_CASES = [
    (2, 64, 8, 9, 32, 3, 1), (3, 128, 16, 20, 8, 3, 1),
    (2, 64, 8, 16, 32, 5, 2), (2, 64, 16, 13, 16, 5, 2), (3, 128, 8, 7, 8, 5, 2), (1, 64, 8, 1, 4, 5, 2),
]


@pytest.mark.parametrize("case", DGRAD_CASES)
def test_conv_dgrad(case):
    """dL/dx of nn.Conv2d: flipped-filter convolution (stride 1) / four parity-class convolutions (stride 2)."""
    lib = emul_lib()
    b, ci, co, h, w, k, s = case
    rs = np.random.RandomState(abs(hash(case)) % 2**31)
    x = rs.randn(b, ci, h, w)
    wt = rs.randn(co, ci, k, k) / np.sqrt(ci * k * k)
    ho, wo = O.conv_out_size(h, k, s, k // 2), O.conv_out_size(w, k, s, k // 2)
    gy = rs.randn(b, co, ho, wo)
    gx_ref, _ = O.conv2d_bwd(x, wt, gy, s, k // 2)
    wsrc, wp = to_aligned(wt.astype(np.float32)), aligned(wt.size)
    if s == 1:
        lib.call("ds_pack_conv_weight_f32", ptr(wsrc), ptr(wp), co, ci, k, 1, None)
    else:
        lib.call("ds_pack_conv_dgrad_s2_f32", ptr(wsrc), ptr(wp), co, ci, None)
    gyh = nhwc(gy.astype(np.float32))
    gx = aligned((b, h, w, ci), fill=np.nan)
    shp = ConvShape(b, h, w, ci, co, k, s)
    lib.call("ds_conv_dgrad_f32", ctypes.byref(shp), ptr(gyh), ptr(wp), ptr(gx), None)
    assert rel_err(nchw(gx), gx_ref) < 2e-6


DGRAD_BF16_CASES = [(2, 64, 64, 9, 32, 3, 1), (2, 64, 64, 16, 32, 5, 2), (2, 64, 16, 13, 16, 5, 2),
                    (3, 128, 64, 7, 8, 5, 2), (1, 64, 32, 1, 4, 5, 2)]


@pytest.mark.parametrize("case", DGRAD_BF16_CASES)
def test_conv_dgrad_bf16x3(case):
    """split-operand bf16 data gradient: stride 1 (flipped bank) and the four stride-2 parity classes
    (zero-padded 3x3 banks, interleaved stores)"""
    lib = emul_lib()
    b, ci, co, h, w, k, s = case
    rs = np.random.RandomState(abs(hash(case)) % 2**31)
    x = rs.randn(b, ci, h, w)
    wt = rs.randn(co, ci, k, k) / np.sqrt(ci * k * k)
    ho, wo = O.conv_out_size(h, k, s, k // 2), O.conv_out_size(w, k, s, k // 2)
    gy = rs.randn(b, co, ho, wo)
    gx_ref, _ = O.conv2d_bwd(x, wt, gy, s, k // 2)
    wsrc = to_aligned(wt.astype(np.float32))
    n = wt.size if s == 1 else 36 * co * ci
    whi, wlo = aligned((n + 1) // 2), aligned((n + 1) // 2)          # bf16 storage in float32 words
    if s == 1:
        lib.call("ds_pack_conv_weight_dgrad_bf16", ptr(wsrc), ptr(whi), ptr(wlo), co, ci, k, None)
    else:
        lib.call("ds_pack_conv_weight_dgrad_s2_bf16", ptr(wsrc), ptr(whi), ptr(wlo), co, ci, None)
    gyh = nhwc(gy.astype(np.float32))
    gx = aligned((b, h, w, ci), fill=np.nan)
    shp = ConvShape(b, h, w, ci, co, k, s)
    lib.call("ds_conv_dgrad_bf16", ctypes.byref(shp), ptr(gyh), ptr(whi), ptr(wlo), ptr(gx), None)
    assert rel_err(nchw(gx), gx_ref) < 3e-5


@pytest.mark.parametrize("C,npix,with_g2,with_act", [(64, 700, True, True), (128, 300, False, True),
                                                       (512, 50, False, False), (256, 1030, True, False)])
def test_bn_bwd(C, npix, with_g2, with_act):
    lib = emul_lib()
    rs = np.random.RandomState(C + npix)
    z = rs.randn(npix, C).astype(np.float32) * 2 + 1
    g1 = rs.randn(npix, C).astype(np.float32)
    g2 = rs.randn(npix, C).astype(np.float32) if with_g2 else None
    act = (rs.rand(npix, C).astype(np.float32) * 30 - 5).clip(0, 20) if with_act else None
    gamma = rs.uniform(0.5, 1.5, C).astype(np.float32)
    mean = z.astype(np.float64).mean(0)
    invstd = 1 / np.sqrt(z.astype(np.float64).var(0) + 1e-5)
    gy_ref = g1.astype(np.float64) + (g2 if with_g2 else 0)
    if with_act:
        gy_ref = gy_ref * ((act > 0) & (act < 20))
    zz = z.astype(np.float64).T.reshape(1, C, npix, 1)
    gz_ref, gg_ref, gb_ref = O.bn_train_bwd(zz, mean, invstd, gamma.astype(np.float64),
                                            gy_ref.T.reshape(1, C, npix, 1))
    rows = lib.raw("ds_bn_bwd_partial_rows")(npix, C)
    bufs = dict(g1=to_aligned(g1), g2=to_aligned(g2) if with_g2 else None, act=to_aligned(act) if with_act else None,
                z=to_aligned(z), mean=to_aligned(mean.astype(np.float32)), invstd=to_aligned(invstd.astype(np.float32)),
                gamma=to_aligned(gamma), gy=aligned((npix, C), fill=np.nan), partial=aligned((rows, C, 2), fill=np.nan),
                coef=aligned(3 * C), gg=aligned(C), gb=aligned(C), gz=aligned((npix, C), fill=np.nan))
    lib.call("ds_bn_bwd_f32", ptr(bufs["g1"]), ptr(bufs["g2"]), ptr(bufs["act"]), ptr(bufs["z"]), ptr(bufs["mean"]),
             ptr(bufs["invstd"]), ptr(bufs["gamma"]), ptr(bufs["gy"]), ptr(bufs["partial"]), ptr(bufs["coef"]),
             ptr(bufs["gg"]), ptr(bufs["gb"]), ptr(bufs["gz"]), npix, C, None)
    assert rel_err(bufs["gy"], gy_ref) < 1e-6
    assert rel_err(bufs["gg"], gg_ref) < 1e-4 and rel_err(bufs["gb"], gb_ref) < 1e-4
    assert rel_err(bufs["gz"], gz_ref[0, :, :, 0].T) < 1e-4



@pytest.mark.parametrize("C,npix,G,with_g2,with_act", [(64, 300, 3, True, True), (512, 40, 3, False, True),
                                                         (128, 77, 2, False, False)])
def test_bn_bwd_group_equals_member_calls(C, npix, G, with_g2, with_act):
    """ds_bn_bwd_group_f32 (all members of a grouped batch in four launches) == one ds_bn_bwd_f32 per member, bit for
    bit, with dgamma / dbeta added in member order"""
    lib = emul_lib()
    rs = np.random.RandomState(C + npix + G)
    z = (rs.randn(G * npix, C) * 2 + 1).astype(np.float32)
    g1 = rs.randn(G * npix, C).astype(np.float32)
    g2 = rs.randn(G * npix, C).astype(np.float32) if with_g2 else None
    act = (rs.rand(G * npix, C).astype(np.float32) * 30 - 5).clip(0, 20) if with_act else None
    gamma = rs.uniform(0.5, 1.5, C).astype(np.float32)
    zs = z.reshape(G, npix, C).astype(np.float64)
    mean = zs.mean(1).astype(np.float32)
    invstd = (1 / np.sqrt(zs.var(1) + 1e-5)).astype(np.float32)
    rows = lib.raw("ds_bn_bwd_partial_rows")(npix, C)
    A = dict(g1=to_aligned(g1), g2=to_aligned(g2) if with_g2 else None, act=to_aligned(act) if with_act else None,
             z=to_aligned(z), mean=to_aligned(mean), invstd=to_aligned(invstd), gamma=to_aligned(gamma))
    # grouped
    gy, gz = aligned((G * npix, C), fill=np.nan), aligned((G * npix, C), fill=np.nan)
    partial, coef = aligned((G, rows, C, 2), fill=np.nan), aligned((G, 3 * C), fill=np.nan)
    msums, gg, gb = aligned((2, G, C), fill=np.nan), aligned(C, fill=np.nan), aligned(C, fill=np.nan)
    lib.call("ds_bn_bwd_group_f32", ptr(A["g1"]), ptr(A["g2"]), ptr(A["act"]), ptr(A["z"]), ptr(A["mean"]),
             ptr(A["invstd"]), ptr(A["gamma"]), ptr(gy), ptr(partial), ptr(coef), ptr(msums), ptr(gg), ptr(gb), ptr(gz),
             npix, C, G, None)
    # one call per member
    gg_sum, gb_sum = np.zeros(C, np.float32), np.zeros(C, np.float32)
    for g in range(G):
        sl = slice(g * npix, (g + 1) * npix)
        m = dict(g1=to_aligned(g1[sl]), g2=to_aligned(g2[sl]) if with_g2 else None,
                 act=to_aligned(act[sl]) if with_act else None, z=to_aligned(z[sl]), mean=to_aligned(mean[g]),
                 invstd=to_aligned(invstd[g]))
        gy1, gz1 = aligned((npix, C), fill=np.nan), aligned((npix, C), fill=np.nan)
        p1, c1, gg1, gb1 = aligned((rows, C, 2), fill=np.nan), aligned(3 * C), aligned(C), aligned(C)
        lib.call("ds_bn_bwd_f32", ptr(m["g1"]), ptr(m["g2"]), ptr(m["act"]), ptr(m["z"]), ptr(m["mean"]), ptr(m["invstd"]),
                 ptr(A["gamma"]), ptr(gy1), ptr(p1), ptr(c1), ptr(gg1), ptr(gb1), ptr(gz1), npix, C, None)
        assert np.array_equal(gy[sl], gy1) and np.array_equal(gz[sl], gz1) and np.array_equal(coef[g], c1)
        gg_sum, gb_sum = gg_sum + gg1, gb_sum + gb1
    assert np.array_equal(gg, gg_sum) and np.array_equal(gb, gb_sum)


WGRAD_CASES = [
    (2, 64, 64, 9, 32, 3, 1),       # stage-1 geometry, ragged rows
    (3, 128, 64, 20, 8, 3, 1),      # multi-row segments
    (5, 64, 128, 10, 4, 3, 1),      # several images per tile
    (2, 64, 128, 16, 32, 5, 2),     # 5x5 stride 2: kernel-row groups, wide-co tiles
    (3, 64, 64, 13, 16, 5, 2),      # 5x5 stride 2, odd height, 64x64 tiles
    (1, 128, 64, 70, 1, 1, 1),      # the fc layer as a 1x1 convolution over [1,B,1,K]
]


@pytest.mark.parametrize("case", WGRAD_CASES)
def test_conv_wgrad(case):
    lib = emul_lib()
    b, ci, co, h, w, k, s = case
    rs = np.random.RandomState(abs(hash(case)) % 2**31)
    x = rs.randn(b, ci, h, w)
    wt = rs.randn(co, ci, k, k)
    ho, wo = O.conv_out_size(h, k, s, k // 2), O.conv_out_size(w, k, s, k // 2)
    gy = rs.randn(b, co, ho, wo)
    _, gw_ref = O.conv2d_bwd(x, wt, gy, s, k // 2, need_gx=False)
    shp = ConvShape(b, h, w, ci, co, k, s)
    n_ws = lib.raw("ds_conv_wgrad_workspace_floats")(ctypes.byref(shp))
    assert n_ws > 0
    ws = aligned(n_ws, fill=np.nan)
    xh, gyh = nhwc(x.astype(np.float32)), nhwc(gy.astype(np.float32))
    gw = aligned((co, ci, k, k), fill=np.nan)
    lib.call("ds_conv_wgrad_f32", ctypes.byref(shp), ptr(xh), ptr(gyh), ptr(ws), ptr(gw), 0, None)
    assert rel_err(gw, gw_ref) < 3e-6


WGRAD_BF16_CASES = [c for c in WGRAD_CASES if c[5] in (3, 5)] + [(4, 128, 128, 10, 4, 5, 2)]


@pytest.mark.parametrize("case", WGRAD_BF16_CASES)
def test_conv_wgrad_bf16x3(case):
    """split-operand bf16 filter gradient (transposing LDS reads) against the float64 oracle"""
    lib = emul_lib()
    b, ci, co, h, w, k, s = case
    rs = np.random.RandomState(abs(hash(case)) % 2**31)
    x = rs.randn(b, ci, h, w)
    wt = rs.randn(co, ci, k, k)
    ho, wo = O.conv_out_size(h, k, s, k // 2), O.conv_out_size(w, k, s, k // 2)
    gy = rs.randn(b, co, ho, wo)
    _, gw_ref = O.conv2d_bwd(x, wt, gy, s, k // 2, need_gx=False)
    shp = ConvShape(b, h, w, ci, co, k, s)
    n_ws = lib.raw("ds_conv_wgrad_bf16_workspace_floats")(ctypes.byref(shp))
    assert n_ws > 0
    ws = aligned(n_ws, fill=np.nan)
    xh, gyh = nhwc(x.astype(np.float32)), nhwc(gy.astype(np.float32))
    gw = aligned((co, ci, k, k), fill=np.nan)
    lib.call("ds_conv_wgrad_bf16", ctypes.byref(shp), ptr(xh), ptr(gyh), ptr(ws), ptr(gw), None)
    assert rel_err(gw, gw_ref) < 3e-5          # 16 mantissa bits per operand


def test_fc_wgrad_feature_permutation():
    lib = emul_lib()
    rs = np.random.RandomState(8)
    B, C, F, N = 40, 32, 4, 64                  # K = C*F = 128
    pooled = rs.randn(B, F * C).astype(np.float32)             # kernel order k' = f*C + c
    gf = rs.randn(B, N).astype(np.float32)
    ref_kp = gf.astype(np.float64).T @ pooled.astype(np.float64)            # [N, f*C+c]
    ref = ref_kp.reshape(N, F, C).transpose(0, 2, 1).reshape(N, C * F)      # reference order c*F+f
    shp = ConvShape(1, B, 1, F * C, N, 1, 1)
    ws = aligned(lib.raw("ds_conv_wgrad_workspace_floats")(ctypes.byref(shp)), fill=np.nan)
    gw = aligned((N, C * F), fill=np.nan)
    pa, ga = to_aligned(pooled), to_aligned(gf)
    lib.call("ds_conv_wgrad_f32", ctypes.byref(shp), ptr(pa), ptr(ga), ptr(ws), ptr(gw), F, None)
    assert rel_err(gw, ref) < 3e-6


@pytest.mark.parametrize("shape", [(2, 160, 64), (3, 13, 64), (1, 7, 20)])
def test_conv1_wgrad(shape):
    lib = emul_lib()
    b, h, w = shape
    rs = np.random.RandomState(h + 1)
    x = rs.randn(b, 1, h, w)
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    gy = rs.randn(b, 64, ho, wo)
    _, gw_ref = O.conv2d_bwd(x, rs.randn(64, 1, 5, 5), gy, 2, 2, need_gx=False)
    shp = ConvShape(b, h, w, 1, 64, 5, 2)
    ws = aligned(lib.raw("ds_conv_wgrad_workspace_floats")(ctypes.byref(shp)), fill=np.nan)
    gw = aligned((64, 1, 5, 5), fill=np.nan)
    xa, ga = to_aligned(x.astype(np.float32)), nhwc(gy.astype(np.float32))
    lib.call("ds_conv_wgrad_f32", ctypes.byref(shp), ptr(xa), ptr(ga), ptr(ws), ptr(gw), 0, None)
    assert rel_err(gw, gw_ref) < 3e-6


@pytest.mark.parametrize("N,M", [(8, 40), (5, 300)])
def test_mine_semihard(N, M):
    lib = emul_lib()
    rs = np.random.RandomState(N)
    a = to_aligned(rs.randn(N, 512).astype(np.float32))
    cand = to_aligned(rs.randn(M, 512).astype(np.float32))
    la = rs.randint(0, 4, N).astype(np.int64)
    lc = rs.randint(0, 4, M).astype(np.int64)
    lc[:3] = la[0]
    d_all = np.sqrt(((a[:, None, :] - cand[None]) ** 2).sum(-1))
    d_p = np.median(d_all, axis=1).astype(np.float32)            # half the candidates are "semi-hard"
    d_p[1] = 1e9                                                 # no semi-hard candidate: falls back to closest
    out = aligned(N, np.int64)
    outd = aligned(N, fill=np.nan)
    la_a, lc_a, dp_a = to_aligned(la, np.int64), to_aligned(lc, np.int64), to_aligned(d_p)
    ws = aligned(lib.raw("ds_mine_workspace_floats")(N, M), fill=np.nan)
    lib.call("ds_mine_semihard_f32", ptr(a), ptr(dp_a), ptr(la_a), ptr(cand), ptr(lc_a), ptr(ws), ptr(out), ptr(outd),
             N, M, 512, None)
    ref = O.mine_semihard(a, d_p, la, cand, lc)
    np.testing.assert_array_equal(out, ref)
    sel = aligned((N, 512), fill=np.nan)
    lib.call("ds_gather_rows_f32", ptr(cand), ptr(out), ptr(sel), N, 512, None)
    np.testing.assert_array_equal(sel, cand[ref])
    # the same rows of three sources in one launch (the refinement window's gather); a negative index gathers zeros
    srcs = [to_aligned(rs.randn(M, 512).astype(np.float32)) for _ in range(3)]
    pick = to_aligned(np.concatenate([ref[:max(1, N - 1)], [-1]])[:N].astype(np.int64), np.int64)
    sel3 = aligned((3, N, 512), fill=np.nan)
    lib.call("ds_gather_rows3_f32", ptr(srcs[0]), ptr(srcs[1]), ptr(srcs[2]), ptr(pick), ptr(sel3), N, 512, None)
    for k in range(3):
        want = np.where(pick[:, None] >= 0, srcs[k][np.maximum(pick, 0)], 0.0).astype(np.float32)
        np.testing.assert_array_equal(sel3[k], want)
    g = to_aligned(rs.randn(N, 512).astype(np.float32))
    idx = to_aligned(np.array([0, 0, 3, 7, 3][:N] + [1] * max(0, N - 5), np.int64), np.int64)
    dst = aligned((M, 512), fill=np.nan)
    lib.call("ds_scatter_add_rows_f32", ptr(g), ptr(idx), ptr(dst), N, M, 512, 0, None)
    ref_s = np.zeros((M, 512), np.float32)
    np.add.at(ref_s, idx, g)
    np.testing.assert_allclose(dst, ref_s, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("N,M,D", [(37, 150, 512), (16, 64, 512), (21, 333, 96), (9, 130, 1024), (40, 70, 2048)])
def test_mine_semihard_shapes_and_ties(N, M, D):
    """ds_mine_semihard_f32 against the oracle: ragged N / M, short and long rows (the anchors per workgroup shrink until
    their rows fit the LDS next to the candidate tile), duplicated candidates (exact distance ties: the lowest index wins),
    an anchor without a semi-hard candidate, an anchor whose every candidate shares its label."""
    lib = emul_lib()
    rs = np.random.RandomState(N * 1000 + M + D)
    a = to_aligned(rs.randn(N, D).astype(np.float32))
    cand = to_aligned(rs.randn(M, D).astype(np.float32))
    cand[M // 2] = cand[3]                                   # exact ties across tiles and inside one
    cand[5] = cand[4]
    la = rs.randint(0, 5, N).astype(np.int64)
    lc = rs.randint(0, 5, M).astype(np.int64)
    la[2] = 99                                               # nobody shares this label: every candidate is "other"
    la[3] = 7
    lc_all_same = lc.copy()
    d_all = np.sqrt(((a[:, None, :] - cand[None]) ** 2).sum(-1))
    # (between two candidates' distances, never ON one: "farther than d_p" must not hinge on the last bit of a sum)
    srt = np.sort(d_all, axis=1)
    d_p = ((srt[:, M // 2] + srt[:, M // 2 + 1]) / 2).astype(np.float32)
    d_p[1] = 1e9                                             # no semi-hard candidate: the closest other-speaker one
    out, outd = aligned(N, np.int64), aligned(N, fill=np.nan)
    ws = aligned(lib.raw("ds_mine_workspace_floats")(N, M), fill=np.nan)
    for labels in (lc, np.full(M, 7, np.int64)):             # second pass: anchor 3 finds no candidate at all (-1)
        la_a, lc_a, dp_a = to_aligned(la, np.int64), to_aligned(labels, np.int64), to_aligned(d_p)
        lib.call("ds_mine_semihard_f32", ptr(a), ptr(dp_a), ptr(la_a), ptr(cand), ptr(lc_a), ptr(ws), ptr(out), ptr(outd),
                 N, M, D, None)
        ref = O.mine_semihard(a, d_p, la, cand, labels)
        np.testing.assert_array_equal(out, ref)
        ok = ref >= 0
        want_d = np.sqrt(((a[ok] - cand[ref[ok]]) ** 2).sum(-1) + 1e-4 / D)
        assert rel_err(outd[ok], want_d.astype(np.float32)) < 2e-6
    assert out[3] == -1


@pytest.mark.parametrize("p", [1.0, 2.0, 3.0, 1.5])
def test_pairwise_distance_any_norm(p):
    """reference model.py:13-18 with self.norm = p (its own call sites pass 2): forward against the oracle, the
    gradient against a central difference of the oracle in float64."""
    lib = emul_lib()
    rs = np.random.RandomState(int(p * 10))
    N, D = 6, 512
    x1 = to_aligned(rs.randn(N, D).astype(np.float32))
    x2 = to_aligned(rs.randn(N, D).astype(np.float32))
    x2[0, :5] = x1[0, :5]                                    # exact zeros of the difference: gradient 0 there
    d = aligned(N, fill=np.nan)
    lib.call("ds_pairwise_distance_p_f32", ptr(x1), ptr(x2), ptr(d), N, D, p, None)
    ref = O.pairwise_distance(x1, x2, p)
    assert rel_err(d, ref) < 2e-6
    if p == 2.0:
        d2 = aligned(N, fill=np.nan)
        lib.call("ds_pairwise_distance_f32", ptr(x1), ptr(x2), ptr(d2), N, D, None)
        assert rel_err(d, d2) < 1e-6
    gd = to_aligned(rs.randn(N).astype(np.float32))
    g1, g2 = aligned((N, D), fill=np.nan), aligned((N, D), fill=np.nan)
    lib.call("ds_pairwise_distance_p_bwd_f32", ptr(x1), ptr(x2), ptr(d), ptr(gd), ptr(g1), ptr(g2), N, D, p, None)
    np.testing.assert_array_equal(g2, -g1)
    assert np.all(g1[0, :5] == 0.0)
    diff = x1.astype(np.float64) - x2.astype(np.float64)
    s = (np.abs(diff) ** p).sum(1) + 1e-4 / D
    want = gd[:, None] * (s ** (1.0 / p - 1.0))[:, None] * np.abs(diff) ** (p - 1.0) * np.sign(diff)
    assert rel_err(g1, want.astype(np.float32)) < 5e-6


BF16_CASES = [
    (2, 16, 64, 9, 32, 3, 1), (3, 32, 128, 20, 8, 3, 1), (5, 16, 128, 10, 4, 3, 1),
    (2, 16, 64, 16, 32, 5, 2), (2, 32, 128, 13, 16, 5, 2), (3, 16, 128, 7, 8, 5, 2),
]


@pytest.mark.parametrize("x3", [True, False])
@pytest.mark.parametrize("case", BF16_CASES)
def test_conv_fwd_bf16(case, x3):
    """bf16 matrix-core variants: bf16x3 (hi/lo split, f32-class accuracy) and plain bf16."""
    lib = emul_lib()
    b, ci, co, h, w, k, s = case
    rs = np.random.RandomState(abs(hash(case)) % 2**31)
    x = rs.randn(b, ci, h, w).astype(np.float32)
    wt = (rs.randn(co, ci, k, k) / np.sqrt(ci * k * k)).astype(np.float32)
    scale = rs.uniform(0.5, 1.5, co).astype(np.float32)
    shift = rs.randn(co).astype(np.float32)
    whi = aligned(wt.size, np.uint16)
    wlo = aligned(wt.size, np.uint16) if x3 else None
    wsrc = to_aligned(wt)
    lib.call("ds_pack_conv_weight_bf16", ptr(wsrc), ptr(whi), ptr(wlo), co, ci, k, None)
    shp = ConvShape(b, h, w, ci, co, k, s)
    ho, wo = O.conv_out_size(h, k, s, k // 2), O.conv_out_size(w, k, s, k // 2)
    rows = lib.raw("ds_conv_bf16_stats_rows")(ctypes.byref(shp), int(x3))
    assert rows > 0
    y = aligned((b, ho, wo, co), fill=np.nan)
    stats = aligned((rows, co, 2), fill=np.nan)
    xh, sc, sh = nhwc(x), to_aligned(scale), to_aligned(shift)
    lib.call("ds_conv_fwd_bf16", ctypes.byref(shp), ptr(xh), ptr(whi), ptr(wlo), ptr(sc), ptr(sh), None, ptr(y),
             ptr(stats), DS_EPI_AFFINE | DS_EPI_STATS, None)
    z = O.conv2d(x.astype(np.float64), wt.astype(np.float64), s, k // 2)
    ref = z * scale[None, :, None, None] + shift[None, :, None, None]
    err = rel_err(nchw(y), ref)
    assert err < (2e-5 if x3 else 2e-2), err
    if not x3:
        assert err > 1e-4                      # really is the reduced-precision path
    tot = stats.astype(np.float64).sum(axis=0)
    np.testing.assert_allclose(tot[:, 0], z.sum(axis=(0, 2, 3)), rtol=2e-2 if not x3 else 1e-4,
                               atol=0.5 if not x3 else 1e-3)


def test_bf16_planner_counts_waves_and_members():
    """Host-side plans of the split-operand bf16 convolution at the training step's sizes (the planner is plain host code:
    the emulated library runs it as is).  A launch with fewer waves than the chip has SIMDs is priced as such -- the 10x4
    layers of ONE 256-utterance member get four-wave workgroups (256 two-wave ones ran 156 us, 256 four-wave ones 128) --
    and the 64-channel 3x3 layers take the four-wave 256x64 tile (484 against 525 us at 768 utterances for the two-wave
    320x64 one).  The fused data-gradient + BatchNorm-backward entry points report which geometries they take for a
    batch of three members: tiles must not straddle members."""
    lib = emul_lib()
    out8 = (ctypes.c_int * 8)()

    def plan(*shape):
        lib.call("ds_conv_bf16_plan_describe", ctypes.byref(ConvShape(*shape)), 1, out8)
        return {"tile": (out8[0], out8[1]), "ni": out8[3], "grid": out8[4], "threads": out8[6]}

    one_member = plan(256, 10, 4, 512, 512, 3, 1)
    assert one_member["threads"] == 256 and one_member["grid"] * (one_member["threads"] // 64) >= 1024, one_member
    batch = plan(768, 10, 4, 512, 512, 3, 1)
    assert batch["grid"] * (batch["threads"] // 64) >= 1024, batch
    for b in (256, 768):
        s1 = plan(b, 80, 32, 64, 64, 3, 1)
        assert s1["tile"] == (256, 64) and s1["threads"] == 256, s1
    rows = lib.raw("ds_conv_dgrad_bnbwd_bf16_rows")
    rows5 = lib.raw("ds_conv_dgrad_s2_bnbwd_bf16_rows")
    for h, w, c in ((80, 32, 64), (40, 16, 128), (20, 8, 256)):
        assert rows(ctypes.byref(ConvShape(768, h, w, c, c, 3, 1)), 3) > 0, (h, w, c)
        assert rows(ctypes.byref(ConvShape(768, h, w, c, c, 3, 1)), 1) > 0
    assert rows5(ctypes.byref(ConvShape(768, 80, 32, 64, 128, 5, 2)), 3) > 0
    # three 40-pixel images per tile straddle members of 256 utterances: the caller falls back to the two-step sequence
    assert rows(ctypes.byref(ConvShape(768, 10, 4, 512, 512, 3, 1)), 3) < 0
    assert rows(ctypes.byref(ConvShape(768, 10, 4, 512, 512, 3, 1)), 1) > 0
    assert rows(ctypes.byref(ConvShape(768, 80, 32, 64, 64, 3, 1)), 5) < 0      # 768 utterances are not 5 members
    assert rows5(ctypes.byref(ConvShape(768, 80, 32, 64, 128, 3, 1)), 3) < 0    # not a 5x5 stride-2 layer


def test_mfma_rate_probe_entry_point():
    """ds_mfma_rate_probe (measurement only: bench.py's live register-only MFMA rate): runs, touches nothing but its sink,
    and reports the floating-point operations of its launch."""
    import ctypes
    lib = emul_lib()
    sink = aligned(4, np.float32, fill=7.0)
    for bf16 in (0, 1):
        flop = ctypes.c_double(0.0)
        assert lib.raw("ds_mfma_rate_probe")(bf16, 1, ptr(sink), ctypes.byref(flop), None) == 0
        assert flop.value > 0 and flop.value % (4 * 4 * 2.0 * 32 * 32 * 16) == 0        # whole workgroups of four waves
    assert (sink == 7.0).all()
    assert lib.raw("ds_mfma_rate_probe")(0, 0, ptr(sink), ctypes.byref(flop), None) == -1
    # operands from real tensors: reads inside the two arrays only (the emulator faults on a stray address)
    a = to_aligned(np.random.RandomState(0).randn(64), np.float16)
    b = to_aligned(np.abs(np.random.RandomState(1).randn(24)), np.float16)
    flop = ctypes.c_double(0.0)
    assert lib.raw("ds_mfma_rate_probe_data")(ptr(a), 64, ptr(b), 24, 1, ptr(sink), ctypes.byref(flop), None) == 0
    assert flop.value > 0 and (sink == 7.0).all()
    assert lib.raw("ds_mfma_rate_probe_data")(ptr(a), 4, ptr(b), 24, 1, ptr(sink), ctypes.byref(flop), None) == -1
