"""The fp16 convolution path on the host emulator (no GPU): the unmodified conv_mfma_f16 sources driven through
the C ABI, compared with the oracle convolution evaluated on the SAME fp16-rounded operands (the kernel's
products are exact in f32, so only the summation order differs).  Covers both buffering modes, both kernel
sizes, ragged tiles, residual / clip epilogues and the f32-output flag."""
import ctypes

import numpy as np
import pytest

import deepspeaker_oracle as O
from conftest import rel_err
from emul_util import aligned, emul_lib, ptr, to_aligned
from deepspeaker_pytorch_amd._native import (ConvShape, DS_CONV_HINT_CHUNK16, DS_CONV_HINT_SINGLE_BUFFER, DS_EPI_AFFINE, DS_EPI_CLIP, DS_EPI_OUT_F16, DS_EPI_OUT_F32,
                                             DS_EPI_RESIDUAL, DS_EPI_STATS)


def nhwc16(x):
    return to_aligned(np.ascontiguousarray(x.transpose(0, 2, 3, 1)).astype(np.float16), np.float16)


def run_conv_f16(lib, x, w, stride, flags=0, scale=None, shift=None, res=None):
    b, ci, h, wd = x.shape
    co, _, k, _ = w.shape
    shp = ConvShape(b, h, wd, ci, co, k, stride)
    ho, wo = O.conv_out_size(h, k, stride, k // 2), O.conv_out_size(wd, k, stride, k // 2)
    out32 = bool(flags & DS_EPI_OUT_F32)
    y = aligned((b, ho, wo, co), np.float32 if out32 else np.float16, fill=np.nan)
    wp, src = aligned(w.size, np.float16), to_aligned(w)
    lib.call("ds_pack_conv_weight_f16", ptr(src), ptr(wp), co, ci, k, None)
    xh = nhwc16(x)
    sc = to_aligned(scale) if scale is not None else None
    sh = to_aligned(shift) if shift is not None else None
    rh = nhwc16(res) if res is not None else None
    lib.call("ds_conv_fwd_f16", ctypes.byref(shp), ptr(xh), ptr(wp), ptr(sc), ptr(sh), ptr(rh), ptr(y), flags, None)
    return np.ascontiguousarray(y.transpose(0, 3, 1, 2)).astype(np.float32)


def describe(lib, case):
    b, ci, co, h, w, k, s = case
    out8 = (ctypes.c_int * 8)()
    lib.call("ds_conv_f16_plan_describe", ctypes.byref(ConvShape(b, h, w, ci, co, k, s)), out8)
    return list(out8)


CASES = [
    # (B, Cin, Cout, H, W, KS, stride)
    (2, 64, 64, 11, 32, 3, 1),               # stage-1 geometry, two chunks, ragged last row block
    (3, 32, 128, 20, 8, 3, 1),               # stage-3 geometry, one chunk, 160-pixel tile = one image
    (5, 96, 128, 10, 4, 3, 1),               # stage-4 geometry: several images per tile, three chunks, ragged tile
    (2, 64, 128, 21, 16, 5, 2),              # 5x5 stride 2, odd height, two chunks
    (3, 32, 256, 9, 8, 5, 2),                # 5x5 s2 into a 5x4 map, multi-image tiles
    (1, 64, 64, 3, 5, 3, 1),                 # tiny map: every tile row ragged
    (1, 64, 128, 21, 64, 5, 2),              # wide stride-2 input: too many staging items -> single-buffered tile
    (1, 32, 64, 12, 100, 3, 1),              # wide 3x3 map (variable-length / wide inputs), single chunk
    (2, 32, 128, 100, 4, 3, 1),              # several row blocks of one image per tile: mixed halo windows (table walk)
]


@pytest.mark.parametrize("case", CASES)
def test_conv_f16_raw(case):
    lib = emul_lib()
    b, ci, co, h, w, k, s = case
    rs = np.random.RandomState(sum(case))
    x = np.abs(rs.randn(b, ci, h, w)).astype(np.float16).astype(np.float32)
    wt = (rs.randn(co, ci, k, k) / np.sqrt(ci * k * k)).astype(np.float16).astype(np.float32)
    y = run_conv_f16(lib, x, wt, s, DS_EPI_OUT_F32)
    ref = O.conv2d(x.astype(np.float64), wt.astype(np.float64), s, k // 2)
    assert y.shape == ref.shape
    assert rel_err(y, ref) < 2e-6, describe(lib, case)


# Larger batches of the bench geometries: on the emulated device (2 "compute units" = 2-4 resident workgroups) every
# persistent workgroup walks many tiles -- top / middle / bottom row blocks, ragged last tiles, several n tiles.
PERSIST_CASES = [
    (3, 64, 64, 27, 32, 3, 1),               # row blocks of one image (linear item offsets), ragged last block
    (9, 32, 128, 20, 8, 3, 1),               # one image per tile
    (11, 64, 128, 10, 4, 3, 1),              # several whole images per tile (item tables), ragged last tile
    (3, 64, 128, 43, 16, 5, 2),              # 5x5 stride 2 row blocks, odd height
    (7, 32, 256, 9, 8, 5, 2),                # 5x5 stride 2, multi-image tiles, two n tiles
    (2, 32, 128, 100, 4, 3, 1),              # a tall narrow map: 32-row blocks
    (7, 64, 256, 10, 4, 3, 1),               # Cout % 256 == 0 on 10x4 maps: the 128 x 256 plan (cfg 7, NSUB = 4), ragged tile
    (8, 32, 512, 10, 4, 3, 1),               # ... two n tiles of 256, one chunk
]


@pytest.mark.parametrize("case", PERSIST_CASES)
@pytest.mark.parametrize("hint", [0, "chunk16"])
def test_conv_f16_persistent_equals_one_tile_kernel(case, hint):
    """conv_mfma_f16_pkernel (persistent workgroups) against conv_mfma_f16_kernel (one tile per workgroup): bit-identical
    raw accumulators, and bit-identical fp16 output of the full epilogue."""
    from deepspeaker_pytorch_amd._native import DS_CONV_HINT_NO_PERSIST
    lib = emul_lib()
    b, ci, co, h, w, k, s = case
    if hint == "chunk16" and k != 5:
        pytest.skip("16-channel chunks exist for the 5x5 layers only")
    hf = DS_CONV_HINT_CHUNK16 if hint else 0
    out8 = (ctypes.c_int * 8)()
    lib.call("ds_conv_f16_plan_describe_hinted", ctypes.byref(ConvShape(b, h, w, ci, co, k, s)), hf, out8)
    taken = out8[7] >= 10000
    assert taken, list(out8)
    wide = co % 256 == 0 and k == 3 and (out8[0], out8[6]) == (128, 128)
    assert (out8[1] == 256) == wide, list(out8)       # the 128-channel-wide register tile where it applies, only there
    rs = np.random.RandomState(17 + sum(case))
    x = (np.abs(rs.randn(b, ci, h, w)) * 2).astype(np.float16).astype(np.float32)
    wt = (rs.randn(co, ci, k, k) / np.sqrt(ci * k * k)).astype(np.float16).astype(np.float32)
    scale = rs.uniform(0.5, 1.5, co).astype(np.float32)
    shift = rs.randn(co).astype(np.float32)
    ho, wo = O.conv_out_size(h, k, s, k // 2), O.conv_out_size(w, k, s, k // 2)
    res = (np.abs(rs.randn(b, co, ho, wo)) * 8).astype(np.float16).astype(np.float32)
    for flags, args in ((DS_EPI_OUT_F32, ()), (DS_EPI_AFFINE | DS_EPI_RESIDUAL | DS_EPI_CLIP, (scale, shift, res)),
                        (DS_EPI_AFFINE | DS_EPI_CLIP, (scale, shift))):
        ya = run_conv_f16(lib, x, wt, s, flags | hf, *args)
        yb = run_conv_f16(lib, x, wt, s, flags | hf | DS_CONV_HINT_NO_PERSIST, *args)
        assert np.isfinite(ya).all() and np.array_equal(ya, yb), (flags, list(out8))
        if wide:                                      # ... and the persistent kernel's own 64-wide form
            from deepspeaker_pytorch_amd._native import DS_CONV_HINT_NO_WIDE
            assert np.array_equal(ya, run_conv_f16(lib, x, wt, s, flags | hf | DS_CONV_HINT_NO_WIDE, *args))
    ref = O.conv2d(x.astype(np.float64), wt.astype(np.float64), s, k // 2)
    assert rel_err(run_conv_f16(lib, x, wt, s, DS_EPI_OUT_F32 | hf), ref) < 2e-6


@pytest.mark.parametrize("case", [CASES[0], CASES[2], CASES[3]])
def test_conv_f16_epilogue(case):
    """affine + residual + clip, fp16 store (round to nearest even of the f32 epilogue value)"""
    lib = emul_lib()
    b, ci, co, h, w, k, s = case
    rs = np.random.RandomState(7 + sum(case))
    x = (np.abs(rs.randn(b, ci, h, w)) * 3).astype(np.float16).astype(np.float32)
    wt = (rs.randn(co, ci, k, k) / np.sqrt(ci * k * k)).astype(np.float16).astype(np.float32)
    scale = rs.uniform(0.5, 1.5, co).astype(np.float32)
    shift = rs.randn(co).astype(np.float32)
    ho, wo = O.conv_out_size(h, k, s, k // 2), O.conv_out_size(w, k, s, k // 2)
    res = (np.abs(rs.randn(b, co, ho, wo)) * 8).astype(np.float16).astype(np.float32)
    flags = DS_EPI_AFFINE | DS_EPI_RESIDUAL | DS_EPI_CLIP
    y = run_conv_f16(lib, x, wt, s, flags, scale, shift, res)
    acc = O.conv2d(x.astype(np.float64), wt.astype(np.float64), s, k // 2)
    ref = np.clip(acc * scale[None, :, None, None] + shift[None, :, None, None] + res, 0.0, 20.0)
    assert (y >= 0).all() and (y <= 20).all()
    # one fp16 rounding of a value in [0, 20]: |err| <= 2^-11 * 16 plus the f32 epilogue's own rounding
    assert np.abs(y - ref).max() <= 20 * 2.0 ** -11 + 1e-5, describe(lib, case)
    assert np.abs(y - ref).mean() < 1.5e-3
    # no residual, no clip: negative values survive
    y2 = run_conv_f16(lib, x, wt, s, DS_EPI_AFFINE | DS_EPI_OUT_F32, scale, shift)
    ref2 = acc * scale[None, :, None, None] + shift[None, :, None, None]
    assert rel_err(y2, ref2) < 2e-6


@pytest.mark.parametrize("case", [CASES[0], CASES[3], CASES[6]])
def test_conv_f16_single_buffered(case):
    """the one-tile variant of the kernel (what the planner falls back to when two tiles do not fit the LDS)"""
    lib = emul_lib()
    b, ci, co, h, w, k, s = case
    rs = np.random.RandomState(11 + sum(case))
    x = np.abs(rs.randn(b, ci, h, w)).astype(np.float16).astype(np.float32)
    wt = (rs.randn(co, ci, k, k) / np.sqrt(ci * k * k)).astype(np.float16).astype(np.float32)
    y = run_conv_f16(lib, x, wt, s, DS_EPI_OUT_F32 | DS_CONV_HINT_SINGLE_BUFFER)
    ref = O.conv2d(x.astype(np.float64), wt.astype(np.float64), s, k // 2)
    assert rel_err(y, ref) < 2e-6


@pytest.mark.parametrize("case", [CASES[3], CASES[4]])
def test_conv_f16_chunk16(case):
    """the 16-channel-chunk variant of the 5x5 kernel (what the planner picks when two 32-channel tiles do not fit)"""
    lib = emul_lib()
    b, ci, co, h, w, k, s = case
    rs = np.random.RandomState(13 + sum(case))
    x = np.abs(rs.randn(b, ci, h, w)).astype(np.float16).astype(np.float32)
    wt = (rs.randn(co, ci, k, k) / np.sqrt(ci * k * k)).astype(np.float16).astype(np.float32)
    y = run_conv_f16(lib, x, wt, s, DS_EPI_OUT_F32 | DS_CONV_HINT_CHUNK16)
    ref = O.conv2d(x.astype(np.float64), wt.astype(np.float64), s, k // 2)
    assert rel_err(y, ref) < 2e-6


@pytest.mark.parametrize("case", [CASES[2], CASES[3], (1, 128, 64, 5, 8, 3, 1)])
def test_conv_f16_split_k(case):
    """small launches: contraction split over workgroups + fixed-order reduction with the epilogue"""
    lib = emul_lib()
    b, ci, co, h, w, k, s = case
    rs = np.random.RandomState(17 + sum(case))
    x = np.abs(rs.randn(b, ci, h, w)).astype(np.float16).astype(np.float32)
    wt = (rs.randn(co, ci, k, k) / np.sqrt(ci * k * k)).astype(np.float16).astype(np.float32)
    scale = rs.uniform(0.5, 1.5, co).astype(np.float32)
    shift = rs.randn(co).astype(np.float32)
    ho, wo = O.conv_out_size(h, k, s, k // 2), O.conv_out_size(w, k, s, k // 2)
    res = (np.abs(rs.randn(b, co, ho, wo)) * 8).astype(np.float16).astype(np.float32)
    shp = ConvShape(b, h, w, ci, co, k, s)
    ws_bytes = lib.raw("ds_conv_f16_splitk_workspace_bytes")(ctypes.byref(shp))
    assert ws_bytes > 0, "these launches are tiny: the contraction must be split"
    ws = aligned(ws_bytes // 4, fill=np.nan)
    wp, src = aligned(wt.size, np.float16), to_aligned(wt)
    lib.call("ds_pack_conv_weight_f16", ptr(src), ptr(wp), co, ci, k, None)
    xh, rh, sc, sh = nhwc16(x), nhwc16(res), to_aligned(scale), to_aligned(shift)
    acc = O.conv2d(x.astype(np.float64), wt.astype(np.float64), s, k // 2)
    y32 = aligned((b, ho, wo, co), fill=np.nan)
    lib.call("ds_conv_fwd_f16_splitk", ctypes.byref(shp), ptr(xh), ptr(wp), None, None, None, ptr(y32), DS_EPI_OUT_F32,
             ptr(ws), ws_bytes, None)
    assert rel_err(y32.transpose(0, 3, 1, 2), acc) < 2e-6
    y16 = aligned((b, ho, wo, co), np.float16, fill=np.nan)
    lib.call("ds_conv_fwd_f16_splitk", ctypes.byref(shp), ptr(xh), ptr(wp), ptr(sc), ptr(sh), ptr(rh), ptr(y16),
             DS_EPI_AFFINE | DS_EPI_RESIDUAL | DS_EPI_CLIP, ptr(ws), ws_bytes, None)
    ref = np.clip(acc * scale[None, :, None, None] + shift[None, :, None, None] + res, 0.0, 20.0)
    assert np.abs(y16.astype(np.float32).transpose(0, 3, 1, 2) - ref).max() <= 20 * 2.0 ** -11 + 1e-5
    # without a workspace the same entry point takes the one-pass path
    y1 = aligned((b, ho, wo, co), fill=np.nan)
    lib.call("ds_conv_fwd_f16_splitk", ctypes.byref(shp), ptr(xh), ptr(wp), None, None, None, ptr(y1), DS_EPI_OUT_F32,
             None, 0, None)
    assert rel_err(y1, y32) < 2e-6


def test_conv_f16_rejects_stats_and_bad_shapes():
    lib = emul_lib()
    x = aligned((1, 4, 4, 32), np.float16, fill=0)
    w = aligned(64 * 32 * 9, np.float16, fill=0)
    y = aligned((1, 4, 4, 64), np.float16, fill=0)
    shp = ConvShape(1, 4, 4, 32, 64, 3, 1)
    fn = lib.raw("ds_conv_fwd_f16")
    assert fn(ctypes.byref(shp), ptr(x), ptr(w), None, None, None, ptr(y), DS_EPI_STATS, None) == -4
    assert fn(ctypes.byref(ConvShape(1, 4, 4, 16, 64, 3, 1)), ptr(x), ptr(w), None, None, None, ptr(y), 0, None) == -1
    assert fn(ctypes.byref(shp), None, ptr(w), None, None, None, ptr(y), 0, None) == -3


def test_conv1_fp16_output():
    """conv1's matrix-core kernel with DS_EPI_OUT_F16 stores the rounded f32 result"""
    lib = emul_lib()
    rs = np.random.RandomState(5)
    b, t = 2, 37
    x = to_aligned(rs.randn(b, t, 64).astype(np.float32))
    w = (rs.randn(64, 1, 5, 5) * 0.2).astype(np.float32)
    wp, src = aligned(w.size), to_aligned(w)
    lib.call("ds_pack_conv1_weight_f32", ptr(src), ptr(wp), 64, None)
    scale = to_aligned(rs.uniform(0.5, 1.5, 64).astype(np.float32))
    shift = to_aligned(rs.randn(64).astype(np.float32))
    ho, wo = (t - 1) // 2 + 1, 32
    y32 = aligned((b, ho, wo, 64), fill=np.nan)
    y16 = aligned((b, ho, wo, 64), np.float16, fill=np.nan)
    fl = DS_EPI_AFFINE | DS_EPI_CLIP
    lib.call("ds_conv5x5s2_c1_fwd_bf16", ptr(x), ptr(wp), ptr(scale), ptr(shift), ptr(y32), None, b, t, 64, 64, fl, None)
    lib.call("ds_conv5x5s2_c1_fwd_bf16", ptr(x), ptr(wp), ptr(scale), ptr(shift), ptr(y16), None, b, t, 64, 64,
             fl | DS_EPI_OUT_F16, None)
    assert np.isfinite(y32).all()
    assert np.array_equal(y16, y32.astype(np.float16))


# the emulated device has 2 "compute units" = 4 resident workgroups (tests/emul/ds_device.h): the 25- and 27-tile
# cases make every persistent workgroup walk 6-7 tiles -- top, middle and bottom row blocks, several images, a ragged
# last block -- and the 9-image case takes the XCD-dealt tile order with idle and unevenly loaded "XCDs"
@pytest.mark.parametrize("geom", [(2, 11, 32, 64), (1, 19, 16, 128), (1, 8, 32, 64), (1, 3, 16, 128), (5, 37, 32, 64),
                                  (9, 20, 16, 128)])
@pytest.mark.parametrize("out", ["f16", "f32", "planes"])
def test_conv_block_fused_equals_two_convolutions(geom, out):
    """ds_conv_block_f16 (both 3x3 layers of a BasicBlock in one kernel, the intermediate in LDS only) must be
    BIT-identical to two ds_conv_fwd_f16 calls: same products, same accumulation order, same fp16 rounding points."""
    from deepspeaker_pytorch_amd._native import DS_EPI_OUT_PLANES16
    lib = emul_lib()
    b, h, w, c = geom
    assert lib.raw("ds_conv_block_f16_supported")(b, h, w, c) == 1
    rs = np.random.RandomState(3 + sum(geom))
    x = (np.abs(rs.randn(b, h, w, c)) * 2).astype(np.float16)
    xa = to_aligned(x, np.float16)
    packs, folds = [], []
    for _ in range(2):
        wt = (rs.randn(c, c, 3, 3) / np.sqrt(c * 9)).astype(np.float32)
        wp, src = aligned(wt.size, np.float16), to_aligned(wt)
        lib.call("ds_pack_conv_weight_f16", ptr(src), ptr(wp), c, c, 3, None)
        packs.append(wp)
        folds.append((to_aligned(rs.uniform(0.5, 1.5, c).astype(np.float32)), to_aligned((rs.randn(c) * 0.5).astype(np.float32))))
    shp = ConvShape(b, h, w, c, c, 3, 1)
    mid = aligned((b, h, w, c), np.float16, fill=np.nan)
    lib.call("ds_conv_fwd_f16", ctypes.byref(shp), ptr(xa), ptr(packs[0]), ptr(folds[0][0]), ptr(folds[0][1]), None, ptr(mid),
             DS_EPI_AFFINE | DS_EPI_CLIP, None)
    oflag = {"f16": 0, "f32": DS_EPI_OUT_F32, "planes": DS_EPI_OUT_PLANES16}[out]
    dt = np.float32 if out == "f32" else np.float16
    ref = aligned((b, h, w, c), dt, fill=np.nan)
    lib.call("ds_conv_fwd_f16", ctypes.byref(shp), ptr(mid), ptr(packs[1]), ptr(folds[1][0]), ptr(folds[1][1]), ptr(xa), ptr(ref),
             DS_EPI_AFFINE | DS_EPI_CLIP | DS_EPI_RESIDUAL | oflag, None)
    got = aligned((b, h, w, c), dt, fill=np.nan)
    lib.call("ds_conv_block_f16", ptr(xa), ptr(packs[0]), ptr(packs[1]), ptr(folds[0][0]), ptr(folds[0][1]), ptr(folds[1][0]),
             ptr(folds[1][1]), ptr(got), b, h, w, c, oflag, None)
    assert np.isfinite(got.astype(np.float32)).all()
    assert np.array_equal(got, ref)


def test_conv_block_xcd_dealt_tile_order(monkeypatch):
    """With >= 8 images and a grid of 8 k workgroups the persistent block kernel deals images to "XCDs" (workgroup w
    serves images b = w % 8 (mod 8)): 9 images on 8 workgroups -- XCD 0 walks two images, the others one."""
    monkeypatch.setenv("DS_EMUL_CUS", "4")
    test_conv_block_fused_equals_two_convolutions((9, 20, 16, 128), "f16")
    test_conv_block_fused_equals_two_convolutions((17, 9, 32, 64), "planes")


@pytest.mark.parametrize("case", [(19, 32, 512, 10, 4, 3, 1), (13, 32, 256, 9, 8, 5, 2)])
def test_conv_f16_persistent_xcd_queues(case, monkeypatch):
    """One tile queue per XCD (grids that are a multiple of 8 whose n-tile count divides 8: workgroup b walks the tiles
    8 j + b % 8, all of one n tile): every tile exactly once whatever the draw order -- bitwise the one-queue walk and the
    one-tile-per-workgroup kernel.  4 emulated CUs = 8 resident two-wave (4 four-wave) workgroups, ragged tile counts."""
    from deepspeaker_pytorch_amd._native import DS_CONV_HINT_NO_PERSIST, DS_CONV_HINT_ONE_QUEUE
    monkeypatch.setenv("DS_EMUL_CUS", "4")
    lib = emul_lib()
    b, ci, co, h, w, k, s = case
    rs = np.random.RandomState(23 + sum(case))
    x = (np.abs(rs.randn(b, ci, h, w)) * 2).astype(np.float16).astype(np.float32)
    wt = (rs.randn(co, ci, k, k) / np.sqrt(ci * k * k)).astype(np.float16).astype(np.float32)
    scale, shift = rs.uniform(0.5, 1.5, co).astype(np.float32), rs.randn(co).astype(np.float32)
    flags = DS_EPI_AFFINE | DS_EPI_CLIP
    ya = run_conv_f16(lib, x, wt, s, flags, scale, shift)
    assert np.isfinite(ya).all()
    assert np.array_equal(ya, run_conv_f16(lib, x, wt, s, flags | DS_CONV_HINT_ONE_QUEUE, scale, shift))
    assert np.array_equal(ya, run_conv_f16(lib, x, wt, s, flags | DS_CONV_HINT_NO_PERSIST, scale, shift))
    ref = O.conv2d(x.astype(np.float64), wt.astype(np.float64), s, k // 2)
    assert rel_err(run_conv_f16(lib, x, wt, s, DS_EPI_OUT_F32), ref) < 2e-6


def test_conv_block_unsupported_geometries():
    lib = emul_lib()
    assert lib.raw("ds_conv_block_f16_supported")(4, 20, 8, 256) == 0
    assert lib.raw("ds_conv_block_f16_supported")(4, 20, 32, 128) == 0
