"""The kernels of the OPT-IN fp16 training step on the host emulator (no GPU): BatchNorm statistics / normalise / backward
over fp16 tensors, the data-gradient banks run through the fp16 convolution (3x3 flipped; 5x5 stride 2 as one 3x3
convolution whose 4 Cin output channels are the parity classes of dX), the fp16 filter-gradient kernels, and the whole
step (Engine level) against the f32 step.  Checkers: the numpy oracle on the SAME fp16-rounded operands."""
import ctypes

import numpy as np
import pytest
import torch

import deepspeaker_oracle as O
from conftest import rel_err
from emul_util import aligned, emul_lib, ptr, to_aligned
from deepspeaker_pytorch_amd._native import ConvShape, DS_EPI_CLIP, DS_EPI_OUT_F32, DS_EPI_RESIDUAL


def r16(a):
    return np.asarray(a, np.float32).astype(np.float16).astype(np.float32)


def nhwc(x, dt=np.float16):
    return to_aligned(np.ascontiguousarray(x.transpose(0, 2, 3, 1)).astype(dt), dt)


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


@pytest.mark.parametrize("G,bm,h,w,c", [(3, 2, 5, 4, 64), (1, 3, 7, 3, 128), (2, 1, 3, 2, 512)])
def test_bn_stats_apply_f16(G, bm, h, w, c):
    lib = emul_lib()
    rs = np.random.RandomState(G * 100 + c)
    z = r16(rs.randn(G * bm, c, h, w) * 3 + 1)
    res = r16(np.abs(rs.randn(G * bm, c, h, w)))
    gamma, beta = rs.uniform(0.5, 1.5, c).astype(np.float32), (rs.randn(c) * 0.1).astype(np.float32)
    rm, rv = (rs.randn(c) * 0.1).astype(np.float32), rs.uniform(0.5, 1.5, c).astype(np.float32)
    n_pix = bm * h * w
    rows = lib.raw("ds_bn_f16_partial_rows")(n_pix, c)
    partial = aligned((G, rows, c, 2), np.float32)
    tables = aligned((4, G, c), np.float32, fill=np.nan)
    zz = nhwc(z)
    g_, b_, rm_, rv_ = (to_aligned(v.copy()) for v in (gamma, beta, rm, rv))
    lib.call("ds_bn_stats_group_f16", ptr(zz), ptr(partial), n_pix, ptr(g_), ptr(b_), 1e-5, 0.1, ptr(rm_), ptr(rv_),
             ptr(tables[0]), ptr(tables[1]), ptr(tables[2]), ptr(tables[3]), c, G, None)
    erm, erv = rm.astype(np.float64), rv.astype(np.float64)
    for m in range(G):                                  # members in call order: three momentum updates
        zm = z[m * bm:(m + 1) * bm].astype(np.float64)
        mean, var = zm.mean(axis=(0, 2, 3)), zm.var(axis=(0, 2, 3))
        np.testing.assert_allclose(tables[0][m], mean, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(tables[1][m], 1 / np.sqrt(var + 1e-5), rtol=1e-5)
        erm = 0.9 * erm + 0.1 * mean
        erv = 0.9 * erv + 0.1 * var * n_pix / max(n_pix - 1, 1)
    np.testing.assert_allclose(rm_, erm, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rv_, erv, rtol=1e-5, atol=1e-6)
    for flags, out_dt in ((DS_EPI_CLIP, np.float16), (DS_EPI_CLIP | DS_EPI_RESIDUAL, np.float16),
                          (DS_EPI_CLIP | DS_EPI_RESIDUAL | DS_EPI_OUT_F32, np.float32), (0, np.float16)):
        y = aligned((G * bm, h, w, c), out_dt, fill=np.nan)
        rr = nhwc(res)
        lib.call("ds_bn_apply_group_f16", ptr(zz), ptr(tables[2]), ptr(tables[3]), ptr(rr), ptr(y), n_pix, c, G, flags, None)
        exp = np.empty_like(z)
        for m in range(G):
            sl = slice(m * bm, (m + 1) * bm)
            exp[sl] = z[sl] * tables[2][m][None, :, None, None] + tables[3][m][None, :, None, None]
        if flags & DS_EPI_RESIDUAL:
            exp = exp + res
        if flags & DS_EPI_CLIP:
            exp = np.clip(exp, 0, 20)
        got = y.transpose(0, 3, 1, 2).astype(np.float32)
        assert np.abs(got - exp).max() <= (1e-5 if out_dt == np.float32 else 1e-2 * max(1.0, np.abs(exp).max()) * 2 ** -3)


@pytest.mark.parametrize("G,bm,h,w,c,parity,act32,with_g2", [(3, 2, 6, 4, 64, False, False, True), (2, 1, 5, 7, 128, True, False, False),
                                                             (1, 2, 4, 4, 256, True, True, True), (3, 1, 3, 3, 64, False, False, False)])
def test_bn_bwd_group_f16(G, bm, h, w, c, parity, act32, with_g2):
    """gy / gz / dgamma / dbeta vs the oracle's BatchNorm + clip backward on the same fp16-rounded inputs, with the
    upstream gradient optionally in the parity-class layout of the stride-2 data gradient (odd map sizes included)"""
    lib = emul_lib()
    rs = np.random.RandomState(7 * G + c + h)
    B = G * bm
    S = 256.0
    z = r16(rs.randn(B, c, h, w) * 2)
    g1 = r16(rs.randn(B, c, h, w) * 1e-3 * S)
    g2 = r16(rs.randn(B, c, h, w) * 1e-3 * S) if with_g2 else None
    act = r16(np.clip(rs.randn(B, c, h, w) * 8 + 8, 0, 20))
    gamma = rs.uniform(0.5, 1.5, c).astype(np.float32)
    mean_t = np.stack([z[m * bm:(m + 1) * bm].mean(axis=(0, 2, 3)) for m in range(G)]).astype(np.float32)
    invstd_t = np.stack([1 / np.sqrt(z[m * bm:(m + 1) * bm].var(axis=(0, 2, 3)) + 1e-5) for m in range(G)]).astype(np.float32)
    n_pix = bm * h * w
    rows = lib.raw("ds_bn_f16_partial_rows")(n_pix, c)
    if parity:          # [B][ceil(h/2)][ceil(w/2)][2][2][c]; cells past an odd edge hold garbage that must not be read
        h2, w2 = (h + 1) // 2, (w + 1) // 2
        g1_dev = aligned((B, h2, w2, 2, 2, c), np.float16, fill=np.nan)
        g1n = g1.transpose(0, 2, 3, 1)
        for hh in range(h):
            for ww in range(w):
                g1_dev[:, hh >> 1, ww >> 1, hh & 1, ww & 1] = g1n[:, hh, ww]
    else:
        g1_dev = nhwc(g1)
    g2_dev = nhwc(g2) if with_g2 else None
    act_dev = nhwc(act, np.float32 if act32 else np.float16)
    gy, gz = aligned((B, h, w, c), np.float16, fill=np.nan), aligned((B, h, w, c), np.float16, fill=np.nan)
    partial, coef = aligned((G, rows, c, 2), np.float32), aligned((G, 3 * c), np.float32)
    gg, gb = aligned(c, np.float32), aligned(c, np.float32)
    mt, it, gm = to_aligned(mean_t), to_aligned(invstd_t), to_aligned(gamma)
    z_dev = nhwc(z)
    lib.call("ds_bn_bwd_group_f16", ptr(g1_dev), int(parity), ptr(g2_dev), ptr(act_dev), int(act32), None, None, ptr(z_dev), ptr(mt),
             ptr(it), ptr(gm), ptr(gy), ptr(partial), ptr(coef), ptr(gg), ptr(gb), ptr(gz), n_pix, h, w, c, G, 1.0 / S, None)
    gsum = g1 + (g2 if with_g2 else 0)
    gy_ref = r16(O.clip_bwd(act, gsum))
    assert np.array_equal(gy.transpose(0, 3, 1, 2).astype(np.float32), gy_ref)
    gg_ref, gb_ref = np.zeros(c), np.zeros(c)
    gz_ref = np.empty_like(z, dtype=np.float64)
    for m in range(G):
        sl = slice(m * bm, (m + 1) * bm)
        gx, g_g, g_b = O.bn_train_bwd(z[sl].astype(np.float64), mean_t[m].astype(np.float64), invstd_t[m].astype(np.float64),
                                      gamma.astype(np.float64), gy_ref[sl].astype(np.float64))
        gz_ref[sl] = gx
        gg_ref += g_g
        gb_ref += g_b
    assert rel_l2(gz.transpose(0, 3, 1, 2), gz_ref) < 1e-3              # fp16 storage of the result
    assert rel_l2(gg, gg_ref / S) < 1e-5 and rel_l2(gb, gb_ref / S) < 1e-5


@pytest.mark.parametrize("G,bm,h,w,c,with_g2,store_gy", [(3, 2, 6, 4, 64, True, True), (2, 2, 5, 3, 128, False, False), (1, 1, 4, 4, 512, False, True)])
def test_bn_bwd_group_f16_mask_from_preactivation(G, bm, h, w, c, with_g2, store_gy):
    """The clip mask re-derived from z and the forward's scale / shift tables must be the mask of the activation
    ds_bn_apply_group_f16 stored (bitwise: same fma, same fp16 rounding) -- so the results equal the act-masked call's --
    also when the masked gradient is never stored (gy NULL: the second launch recomputes it)."""
    lib = emul_lib()
    rs = np.random.RandomState(11 * G + c)
    B = G * bm
    S = 256.0
    z = r16(rs.randn(B, c, h, w) * 4 + 1)
    g1 = r16(rs.randn(B, c, h, w) * 1e-3 * S)
    g2 = r16(rs.randn(B, c, h, w) * 1e-3 * S) if with_g2 else None
    gamma = rs.uniform(0.5, 1.5, c).astype(np.float32)
    mean_t = to_aligned(np.stack([z[m * bm:(m + 1) * bm].mean(axis=(0, 2, 3)) for m in range(G)]).astype(np.float32))
    invstd_t = to_aligned(np.stack([1 / np.sqrt(z[m * bm:(m + 1) * bm].var(axis=(0, 2, 3)) + 1e-5) for m in range(G)]).astype(np.float32))
    # tables that put plenty of values on both clip boundaries (and exactly on fp16 ties)
    sc_t = to_aligned((rs.uniform(2.0, 6.0, (G, c))).astype(np.float32))
    sh_t = to_aligned((rs.randn(G, c) * 3 + 6).astype(np.float32))
    n_pix = bm * h * w
    rows = lib.raw("ds_bn_f16_partial_rows")(n_pix, c)
    z_dev, g1_dev, g2_dev = nhwc(z), nhwc(g1), (nhwc(g2) if with_g2 else None)
    act = aligned((B, h, w, c), np.float16, fill=np.nan)            # what the forward stored
    lib.call("ds_bn_apply_group_f16", ptr(z_dev), ptr(sc_t), ptr(sh_t), None, ptr(act), n_pix, c, G, DS_EPI_CLIP, None)
    frac = float(((act > 0) & (act < 20)).mean())
    assert 0.2 < frac < 0.9, frac
    gm = to_aligned(gamma)
    outs = []
    for maskz in (False, True):
        gy = aligned((B, h, w, c), np.float16, fill=np.nan) if (store_gy or not maskz) else None
        gz = aligned((B, h, w, c), np.float16, fill=np.nan)
        partial, coef = aligned((G, rows, c, 2), np.float32), aligned((G, 3 * c), np.float32)
        gg, gb = aligned(c, np.float32), aligned(c, np.float32)
        lib.call("ds_bn_bwd_group_f16", ptr(g1_dev), 0, ptr(g2_dev), None if maskz else ptr(act), 0,
                 ptr(sc_t) if maskz else None, ptr(sh_t) if maskz else None, ptr(z_dev), ptr(mean_t), ptr(invstd_t), ptr(gm),
                 ptr(gy), ptr(partial), ptr(coef), ptr(gg), ptr(gb), ptr(gz), n_pix, h, w, c, G, 1.0 / S, None)
        outs.append((gy, gz.copy(), gg.copy(), gb.copy()))
    (gy0, gz0, gg0, gb0), (gy1, gz1, gg1, gb1) = outs
    if gy1 is not None:
        assert np.array_equal(gy0, gy1)
    assert np.array_equal(gz0, gz1) and np.array_equal(gg0, gg1) and np.array_equal(gb0, gb1)
    if with_g2 or not store_gy:     # combinations the ABI refuses: no stored gradient with g2 / parity / act masks
        gz = aligned((B, h, w, c), np.float16)
        partial, coef = aligned((G, rows, c, 2), np.float32), aligned((G, 3 * c), np.float32)
        gg, gb = aligned(c, np.float32), aligned(c, np.float32)
        rc = lib.raw("ds_bn_bwd_group_f16")(ptr(g1_dev), 0, ptr(g1_dev), None, 0, ptr(sc_t), ptr(sh_t), ptr(z_dev), ptr(mean_t),
                                            ptr(invstd_t), ptr(gm), None, ptr(partial), ptr(coef), ptr(gg), ptr(gb), ptr(gz),
                                            n_pix, h, w, c, G, 1.0 / S, None)
        assert rc != 0


def conv16(lib, x_nhwc16, bank, shp, out_c):
    ho, wo = O.conv_out_size(shp.H, shp.KS, shp.stride, shp.KS // 2), O.conv_out_size(shp.W, shp.KS, shp.stride, shp.KS // 2)
    y = aligned((shp.B, ho, wo, out_c), np.float16, fill=np.nan)
    lib.call("ds_conv_fwd_f16", ctypes.byref(shp), ptr(x_nhwc16), ptr(bank), None, None, None, ptr(y), 0, None)
    return y


@pytest.mark.parametrize("b,ci,co,h,w", [(2, 64, 64, 6, 8), (3, 128, 64, 5, 4)])
def test_dgrad_3x3_through_the_fp16_convolution(b, ci, co, h, w):
    lib = emul_lib()
    rs = np.random.RandomState(ci + h)
    wt = r16(rs.randn(co, ci, 3, 3) / np.sqrt(ci * 9))
    gy = r16(rs.randn(b, co, h, w))
    bank = aligned(wt.size, np.float16)
    wsrc, gsrc = to_aligned(wt), nhwc(gy)                   # (named: a temporary would be freed before the call reads it)
    lib.call("ds_pack_conv_weight_dgrad_f16", ptr(wsrc), ptr(bank), co, ci, 3, 1, None)
    got = conv16(lib, gsrc, bank, ConvShape(b, h, w, co, ci, 3, 1), ci).transpose(0, 3, 1, 2).astype(np.float32)
    gx, _ = O.conv2d_bwd(np.zeros((b, ci, h, w)), wt.astype(np.float64), gy.astype(np.float64), 1, 1)
    assert rel_l2(got, gx) < 1e-3


@pytest.mark.parametrize("b,ci,co,h,w", [(2, 64, 128, 8, 6), (1, 64, 64, 7, 5), (2, 128, 256, 5, 4)])
def test_dgrad_5x5_stride2_as_one_3x3_convolution_with_parity_classes(b, ci, co, h, w):
    """dX of a 5x5 stride-2 pad-2 layer with input [h, w] (even and odd sizes): the parity bank through ds_conv_fwd_f16
    gives [B][ho][wo][2][2][ci]; class (a, b) of cell (i, j) is dX[2i+a, 2j+b]"""
    lib = emul_lib()
    rs = np.random.RandomState(ci + co + h)
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    wt = r16(rs.randn(co, ci, 5, 5) / np.sqrt(ci * 25))
    gy = r16(rs.randn(b, co, ho, wo))
    bank = aligned(36 * co * ci, np.float16)
    wsrc, gsrc = to_aligned(wt), nhwc(gy)
    lib.call("ds_pack_conv_weight_dgrad_f16", ptr(wsrc), ptr(bank), co, ci, 5, 2, None)
    out = conv16(lib, gsrc, bank, ConvShape(b, ho, wo, co, 4 * ci, 3, 1), 4 * ci).astype(np.float32)
    out = out.reshape(b, ho, wo, 2, 2, ci)
    gx, _ = O.conv2d_bwd(np.zeros((b, ci, h, w)), wt.astype(np.float64), gy.astype(np.float64), 2, 2)
    got = np.zeros((b, ci, h, w), np.float32)
    for hh in range(h):
        for ww in range(w):
            got[:, :, hh, ww] = out[:, hh >> 1, ww >> 1, hh & 1, ww & 1]
    assert rel_l2(got, gx) < 1e-3


@pytest.mark.parametrize("b,ci,co,h,w,ks,st", [(2, 64, 64, 9, 8, 3, 1), (3, 64, 128, 10, 4, 3, 1), (2, 64, 128, 11, 8, 5, 2),
                                               (1, 128, 64, 6, 16, 5, 2), (5, 64, 64, 40, 16, 3, 1)])
def test_wgrad_f16(b, ci, co, h, w, ks, st):
    lib = emul_lib()
    rs = np.random.RandomState(ci + co + h + ks)
    pad = ks // 2
    ho, wo = O.conv_out_size(h, ks, st, pad), O.conv_out_size(w, ks, st, pad)
    x = r16(np.abs(rs.randn(b, ci, h, w)))
    S = 512.0
    gy = r16(rs.randn(b, co, ho, wo) * 1e-3 * S)
    shp = ConvShape(b, h, w, ci, co, ks, st)
    n_ws = lib.raw("ds_conv_wgrad_f16_workspace_floats")(ctypes.byref(shp))
    assert n_ws > 0
    ws, gw = aligned(n_ws, np.float32), aligned((co, ci, ks, ks), np.float32, fill=np.nan)
    x_dev, g_dev = nhwc(x), nhwc(gy)
    lib.call("ds_conv_wgrad_f16", ctypes.byref(shp), ptr(x_dev), ptr(g_dev), ptr(ws), ptr(gw), 1.0 / S, None)
    _, ref = O.conv2d_bwd(x.astype(np.float64), np.zeros((co, ci, ks, ks)), gy.astype(np.float64) / S, st, pad, need_gx=False)
    assert rel_l2(gw, ref) < 2e-6


def test_wgrad_c1_f16():
    lib = emul_lib()
    rs = np.random.RandomState(3)
    b, h, w = 3, 21, 16
    ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
    x = rs.randn(b, 1, h, w).astype(np.float32)
    S = 1024.0
    gy = r16(rs.randn(b, 64, ho, wo) * 1e-3 * S)
    shp = ConvShape(b, h, w, 1, 64, 5, 2)
    ws = aligned(lib.raw("ds_conv_wgrad_workspace_floats")(ctypes.byref(shp)), np.float32)
    gw = aligned((64, 1, 5, 5), np.float32, fill=np.nan)
    x_dev, g_dev = to_aligned(x.reshape(b, h, w)), nhwc(gy)
    lib.call("ds_conv_wgrad_c1_f16", ctypes.byref(shp), ptr(x_dev), ptr(g_dev), ptr(ws), ptr(gw), 1.0 / S, None)
    _, ref = O.conv2d_bwd(x.astype(np.float64), np.zeros((64, 1, 5, 5)), gy.astype(np.float64) / S, 2, 2, need_gx=False)
    assert rel_l2(gw, ref) < 1e-5


def test_scale_cast():
    lib = emul_lib()
    x = to_aligned(np.random.RandomState(1).randn(64).astype(np.float32) * 1e-4)
    y = aligned(64, np.float16)
    lib.call("ds_scale_cast_f32_to_f16", ptr(x), ptr(y), 64, 1024.0, None)
    assert np.array_equal(y, (x * np.float32(1024.0)).astype(np.float16))


@pytest.mark.parametrize("G", [1, 3])
def test_whole_step_fp16_vs_masked_oracle(G):
    """Engine level: forward_train_group_f16 + backward_train_f16 (2 stages, tiny batch) against the torch restatement of
    the step evaluated in float64 WITH THE fp16 FORWARD'S OWN clipped-ReLU masks (both sides then differentiate the same
    piecewise-linear function; an unmasked comparison would measure how many of a tiny map's masks fp16 rounding flips --
    percents -- not the kernels): embeddings, running statistics, every parameter gradient."""
    import torch_restatement as TR
    from deepspeaker_pytorch_amd.engine import BNParams, Engine
    from deepspeaker_pytorch_amd.train_f16 import backward_train_f16, forward_train_group_f16
    eng = Engine(emul_lib())
    n_stages = 2
    sd = O.make_state_dict(seed=71, num_classes=4, n_stages=n_stages)
    bm = 2
    xs = [torch.from_numpy(O.make_input(seed=80 + g, batch=bm, frames=16)) for g in range(G)]
    names = []
    for i in range(1, n_stages + 1):
        names += [f"model.bn{i}", f"model.layer{i}.0.bn1", f"model.layer{i}.0.bn2"]
    tsd = {k: torch.from_numpy(np.array(v)) for k, v in sd.items()}
    bns = {n: BNParams(tsd[n + ".weight"], tsd[n + ".bias"], tsd[n + ".running_mean"].clone(), tsd[n + ".running_var"].clone())
           for n in names}
    pw = eng.pack_weights(tsd, n_stages, with_dgrad=True, with_f16=True, f32_banks=False, with_f16_dgrad=True)
    embs, saved = forward_train_group_f16(eng, xs, pw, bns, save=True)
    e = torch.cat([t.clone() for t in embs])
    ge = torch.from_numpy(np.random.RandomState(5).randn(*e.shape).astype(np.float32) * 1e-2)
    grads = backward_train_f16(eng, {n: b.weight for n, b in bns.items()}, pw, saved, ge, loss_scale=1024.0)
    masks = []
    for g in range(G):
        d = {}
        for key, act in saved.acts.items():
            a = act[g * bm:(g + 1) * bm].float()
            d[key] = ((a > 0) & (a < 20)).permute(0, 3, 1, 2).contiguous()      # the backward kernels' rule
        masks.append(d)
    ref = TR.triplet_train_step(tsd, xs, masks=masks, dtype=torch.float64, n_stages=n_stages,
                                ge=[ge[g * bm:(g + 1) * bm] for g in range(G)])
    assert rel_err(e.numpy(), torch.cat(ref["embeddings"]).float().numpy()) < 3e-3
    for n in names:
        np.testing.assert_allclose(bns[n].running_mean.numpy(), ref["running"][n + ".running_mean"].float().numpy(), rtol=2e-3, atol=2e-3)
        np.testing.assert_allclose(bns[n].running_var.numpy(), ref["running"][n + ".running_var"].float().numpy(), rtol=5e-3, atol=2e-3)
    assert set(grads) == set(ref["grads"])
    worst = {k: rel_l2(grads[k].numpy(), ref["grads"][k].numpy()) for k in grads}
    print(sorted(worst.items(), key=lambda kv: -kv[1])[:4])
    assert max(worst.values()) < 6e-3, worst


def test_overflowing_loss_scale_raises_the_flag_and_the_optimizer_skips_the_step():
    """the fp16 step's loss scale is static and the scaled cast does not saturate.  A scale that pushes dL/dz
    past fp16's 65504 must end the pass with the overflow flag set (inf / NaN in the filter gradients), the fused
    optimizers given that flag must leave parameters AND state untouched, and a sane scale must leave the flag clear."""
    from deepspeaker_pytorch_amd.engine import BNParams, Engine
    from deepspeaker_pytorch_amd.optim import FusedAdagrad, FusedAdam, FusedSGD
    from deepspeaker_pytorch_amd.train_f16 import backward_train_f16, forward_train_group_f16
    lib = emul_lib()
    eng = Engine(lib)
    n_stages = 2
    sd = O.make_state_dict(seed=72, num_classes=4, n_stages=n_stages)
    xs = [torch.from_numpy(O.make_input(seed=90, batch=2, frames=16))]
    names = []
    for i in range(1, n_stages + 1):
        names += [f"model.bn{i}", f"model.layer{i}.0.bn1", f"model.layer{i}.0.bn2"]
    tsd = {k: torch.from_numpy(np.array(v)) for k, v in sd.items()}
    bns = {n: BNParams(tsd[n + ".weight"], tsd[n + ".bias"], tsd[n + ".running_mean"].clone(), tsd[n + ".running_var"].clone())
           for n in names}
    pw = eng.pack_weights(tsd, n_stages, with_dgrad=True, with_f16=True, f32_banks=False, with_f16_dgrad=True)
    ge = torch.from_numpy(np.random.RandomState(6).randn(2, 512).astype(np.float32) * 1e-2)
    flag = torch.zeros(1, dtype=torch.int32)
    # passes only RAISE the flag (a step may be several backward passes -- three `model(x)` calls, gradient
    # accumulation -- and a clean later pass must not hide an earlier overflow): clean -> 0, overflow -> 1, clean -> still 1
    for scale, expect in ((1024.0, 0), (1e9, 1), (1024.0, 1)):
        embs, saved = forward_train_group_f16(eng, xs, pw, bns, save=True)
        grads = backward_train_f16(eng, {n: b.weight for n, b in bns.items()}, pw, saved, ge, loss_scale=scale,
                                   overflow_flag=flag)
        assert int(flag) == expect, (scale, int(flag))
        finite = all(bool(torch.isfinite(g).all()) for k, g in grads.items() if "conv" in k or "fc" in k)
        assert finite == (scale == 1024.0)
    # the flag is still 1: every fused optimizer leaves parameter and state alone and CONSUMES the flag; cleared, they step
    for cls, kw in ((FusedAdagrad, dict(lr=0.1)), (FusedSGD, dict(lr=0.1, momentum=0.9, dampening=0.9)), (FusedAdam, dict(lr=0.1))):
        p = torch.nn.Parameter(torch.arange(40, dtype=torch.float32).reshape(5, 8).clone())
        p.grad = torch.ones_like(p)
        opt = cls([p], **kw)
        opt._engine = eng
        opt.skip_flag = flag
        before = p.detach().clone()
        opt.step()
        assert torch.equal(p.detach(), before), cls.__name__
        for v in opt.state[p].values():
            if torch.is_tensor(v) and v.numel() > 1:
                assert float(v.abs().sum()) == 0.0, cls.__name__
        assert int(flag) == 0, cls.__name__             # consumed by the step that honoured it
        opt.step()
        assert not torch.equal(p.detach(), before), cls.__name__
        flag.fill_(1)
    # what create_optimizer wires: the flag resolved through the MODEL at every step (a weak reference), consumed by the
    # step and latched for grad_overflow / update_loss_scale
    from deepspeaker_pytorch_amd import model as M
    from deepspeaker_pytorch_amd.optim import create_optimizer
    mdl = M.DeepSpeakerModel(512, 4, n_stages=2, precision="f16", train_precision="f16")
    opt = create_optimizer(mdl, 0.1, "adagrad")
    opt._engine = eng
    for p in mdl.parameters():
        p.grad = torch.ones_like(p)
    mflag = mdl.grad_overflow_flag()
    assert opt.skip_flag.data_ptr() == mflag.data_ptr()
    before = {n: p.detach().clone() for n, p in mdl.named_parameters()}
    mflag.fill_(1)
    opt.step()
    assert all(torch.equal(p.detach(), before[n]) for n, p in mdl.named_parameters())
    assert int(mflag) == 0 and mdl.grad_overflow and mdl.update_loss_scale() and mdl.loss_scale == 512.0
    opt.step()
    assert not any(torch.equal(p.detach(), before[n]) for n, p in mdl.named_parameters())
    assert not mdl.grad_overflow and not mdl.update_loss_scale()
    # the detector itself: tail elements, NaN and -inf
    x = to_aligned(np.zeros(4099, np.float32))
    f = aligned(1, np.int32, fill=0)
    lib.call("ds_nonfinite_flag_f32", ptr(x), 4099, ptr(f), None)
    assert int(f[0]) == 0
    for pos, val in ((4098, np.nan), (5, -np.inf), (4096, np.inf)):
        x[:] = 0
        x[pos] = val
        f[0] = 0
        lib.call("ds_nonfinite_flag_f32", ptr(x), 4099, ptr(f), None)
        assert int(f[0]) == 1, (pos, val)
