"""Parity of the HIP path (through the C ABI, on a real MI355X) with the oracle and with the
recorded outputs of the unmodified reference.  Tolerances: north_star asks for 1e-3 relative on
embeddings and loss and identical triplet selections; the exact-f32 MFMA path is held to 2e-5."""
import ctypes
import os

import numpy as np
import pytest
import torch

import deepspeaker_oracle as O
from conftest import rel_err

pytestmark = pytest.mark.gpu

EMB_TOL = 2e-5          # exact-f32 MFMA path vs reference fp32 / oracle fp64 (north_star bar: 1e-3)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "-m gpu tests need an MI355X"
    return torch.device("cuda", 0)


@pytest.fixture(autouse=True)
def _restore_engine_switches():
    yield
    from deepspeaker_pytorch_amd.engine import Engine
    Engine.MEMBER_STREAMS = True


def build_model(sd, n_stages=4, num_classes=16):
    from deepspeaker_pytorch_amd.model import DeepSpeakerModel
    m = DeepSpeakerModel(512, num_classes, n_stages=n_stages)
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    return m.cuda()


def test_native_library_is_the_hip_one(dev):
    from deepspeaker_pytorch_amd import _native
    lib = _native.load()
    assert os.path.basename(lib.path) == "libdeepspeaker_hip.so"
    with open("/proc/self/maps") as f:
        assert "libdeepspeaker_hip.so" in f.read()


def test_cpu_tensors_are_refused(dev):
    from deepspeaker_pytorch_amd.model import PairwiseDistance, TripletMarginLoss
    m = build_model(O.make_state_dict(seed=1, num_classes=16)).eval()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 1, 160, 64))
    with pytest.raises(RuntimeError):
        PairwiseDistance(2).forward(torch.zeros(2, 512), torch.zeros(2, 512))
    with pytest.raises(RuntimeError):
        TripletMarginLoss(0.1).forward(torch.zeros(2, 512), torch.zeros(2, 512), torch.zeros(2, 512))


def test_full_eval_vs_reference_golden(dev, golden):
    sd = O.make_state_dict(seed=11, num_classes=16)
    m = build_model(sd).eval()
    x = O.make_input(seed=12, batch=6)
    from deepspeaker_pytorch_amd.model import get_engine
    taps = {}
    with torch.no_grad():
        e = m(torch.from_numpy(x).cuda())
        get_engine().forward_eval(torch.from_numpy(x).cuda(), m._packed(), m._folded(), taps)
        cls = m.forward_classifier(torch.from_numpy(x).cuda())
    e = e.cpu().numpy()
    assert rel_err(e, golden["full_eval_emb"]) < EMB_TOL
    otaps = {}
    ref64 = O.forward(sd, x, dtype=np.float64, taps=otaps)
    assert rel_err(e, ref64) < EMB_TOL
    for k, v in taps.items():
        assert rel_err(v.cpu().numpy().transpose(0, 3, 1, 2), otaps[k]) < EMB_TOL, k
    assert rel_err(taps["stage1.a"].cpu().numpy().transpose(0, 3, 1, 2)[:1, :, :16], golden["full_eval_stage1_a"]) < EMB_TOL
    assert rel_err(cls.cpu().numpy(), golden["full_eval_cls"]) < 1e-4
    assert m.features is not None and m.features.shape == (6, 512)


@pytest.mark.parametrize("T", [100, 237, 402])
def test_variable_length(dev, golden, T):
    sd = O.make_state_dict(seed=11, num_classes=16)
    m = build_model(sd).eval()
    x = O.make_input(seed=100 + T, batch=2, frames=T)
    with torch.no_grad():
        e = m(torch.from_numpy(x).cuda()).cpu().numpy()
    assert rel_err(e, golden[f"full_eval_T{T}_emb"]) < EMB_TOL


def test_small_model_config0(dev, golden):
    """BASELINE configs[0]: ResCNN-small (64/128 ch, 2 res-blocks), 32 random 64x160 utterances."""
    sd = O.make_state_dict(seed=21, num_classes=16, n_stages=2)
    m = build_model(sd, n_stages=2).eval()
    x = O.make_input(seed=22, batch=32)
    with torch.no_grad():
        e = m(torch.from_numpy(x).cuda()).cpu().numpy()
    assert rel_err(e, golden["small_eval_emb"]) < EMB_TOL


def test_train_forward_and_running_stats(dev, golden):
    sd = O.make_state_dict(seed=31, num_classes=16)
    m = build_model(sd).train()
    xs = [O.make_input(seed=32 + i, batch=8) for i in range(3)]
    with torch.no_grad():
        embs = [m(torch.from_numpy(x).cuda()) for x in xs]
    for e, k in zip(embs, "apn"):
        assert rel_err(e.cpu().numpy(), golden[f"full_train_emb_{k}"]) < 5e-5
    msd = m.state_dict()
    for k in golden.files:
        if k.startswith("full_train_stat/"):
            name = k.split("/", 1)[1]
            if name.endswith("num_batches_tracked"):
                assert int(msd[name]) == 3
            else:
                assert rel_err(msd[name].cpu().numpy(), golden[k]) < 5e-5, name
    from deepspeaker_pytorch_amd.model import TripletMarginLoss
    from deepspeaker_pytorch_amd.mining import select_triplets
    loss = TripletMarginLoss(0.1).forward(*embs)
    ref = float(golden["full_train_loss"])
    assert abs(float(loss) - ref) <= 1e-3 * max(abs(ref), 1e-6)
    sel = select_triplets(*embs, margin=0.1)
    gap = np.abs(golden["full_train_d_n"] - golden["full_train_d_p"] - 0.1).min()
    print("min |d_n - d_p - margin| =", gap)
    np.testing.assert_array_equal(sel.indices.cpu().numpy(), golden["full_train_selected"])


def test_loss_side_vs_reference(dev, golden):
    from deepspeaker_pytorch_amd.model import PairwiseDistance, TripletMarginLoss
    from deepspeaker_pytorch_amd.mining import select_triplets
    rs = np.random.RandomState(41)
    N = 96
    base = rs.randn(N, 512).astype(np.float32)
    a = (base / np.linalg.norm(base, axis=1, keepdims=True) * 10).astype(np.float32)
    p = a + rs.randn(N, 512).astype(np.float32) * 0.05
    n = a + rs.randn(N, 512).astype(np.float32) * 0.05
    ta, tp, tn = (torch.from_numpy(v).cuda().requires_grad_(True) for v in (a, p, n))
    pd = PairwiseDistance(2)
    assert rel_err(pd.forward(ta, tp).detach().cpu().numpy(), golden["loss_d_p"]) < 1e-6
    loss = TripletMarginLoss(0.1).forward(ta, tp, tn)
    assert abs(float(loss) - float(golden["loss_value"])) < 1e-6
    loss.backward()
    assert rel_err(ta.grad.cpu().numpy()[:16], golden["loss_grad_a"]) < 1e-5
    assert rel_err(tp.grad.cpu().numpy()[:16], golden["loss_grad_p"]) < 1e-5
    assert rel_err(tn.grad.cpu().numpy()[:16], golden["loss_grad_n"]) < 1e-5
    sel = select_triplets(ta, tp, tn, margin=0.1)
    np.testing.assert_array_equal(sel.indices.cpu().numpy(), golden["loss_selected"])
    assert sel.n_correct == int(golden["loss_n_correct"])
    assert abs(float(sel.mean_diff) - float(golden["loss_mean_diff"])) < 1e-6
    # PairwiseDistance backward
    tb = torch.from_numpy(p).cuda().requires_grad_(True)
    d = pd.forward(ta.detach().requires_grad_(True), tb)
    d.sum().backward()
    g_ref = -(a - p) / golden["loss_d_p"][:, None]
    assert rel_err(tb.grad.cpu().numpy(), g_ref) < 1e-5
    # test(): mean over 8 crop-pair distances (train_triplet.py:348-350)
    scores = pd.forward(ta, tp).detach().reshape(N // 8, 8).mean(dim=1)
    assert rel_err(scores.cpu().numpy(), golden["loss_test_scores"]) < 1e-6


def test_full_size_properties(dev):
    """BASELINE configs[1] size (B = 256): size-independent checks -- unit norm x10, bitwise run-to-run
    determinism, and batch-composition invariance of eval-mode embeddings."""
    sd = O.make_state_dict(seed=0, num_classes=16)
    m = build_model(sd).eval()
    g = torch.Generator().manual_seed(7)
    x = torch.randn(256, 1, 160, 64, generator=g).cuda()
    with torch.no_grad():
        e1 = m(x).clone()
        e2 = m(x).clone()
        e_small = m(x[40:46].contiguous()).clone()
    assert torch.isfinite(e1).all()
    assert torch.equal(e1, e2)
    nrm = e1.double().norm(dim=1)
    assert float((nrm - 10).abs().max()) < 1e-4
    assert rel_err(e1[40:46].cpu().numpy(), e_small.cpu().numpy()) < 1e-6
    ref = O.forward(sd, x[:4].cpu().numpy(), dtype=np.float64)
    assert rel_err(e1[:4].cpu().numpy(), ref) < EMB_TOL


def test_bench_size_properties_bf16x3(dev):
    """The bench.py step at its full size (768 utterances, default bf16x3 arithmetic): size-independent
    properties -- bitwise determinism, unit norm x10, batch-composition invariance, a small slice against
    the float64 oracle, the filter's indices == ascending where(mask) of the distances it reports, and the
    semi-hard search's contract (other speaker; farther than the positive when such a candidate exists)."""
    from deepspeaker_pytorch_amd.mining import mine_semihard_negatives, select_triplets
    from deepspeaker_pytorch_amd.model import DeepSpeakerModel, PairwiseDistance
    sd = O.make_state_dict(seed=0, num_classes=16)
    m = DeepSpeakerModel(512, 16, precision="bf16x3")
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    m = m.cuda().eval()
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(768, 1, 160, 64, generator=g).cuda()
    with torch.no_grad():
        e1 = m(x).clone()
        e2 = m(x).clone()
        e_part = m(x[256:512].contiguous()).clone()
    assert torch.isfinite(e1).all() and torch.equal(e1, e2)
    assert float((e1.double().norm(dim=1) - 10).abs().max()) < 1e-4
    assert torch.equal(e1[256:512], e_part)             # the contraction order of a pixel does not depend on tiling
    ref = O.forward(sd, x[764:768].cpu().numpy(), dtype=np.float64)
    assert rel_err(e1[764:768].cpu().numpy(), ref) < 3e-5
    a, p, n = e1[:256], e1[256:512], e1[512:]
    sel = select_triplets(a, p, n, margin=0.1)
    d_p = PairwiseDistance(2).forward(a, p).cpu().numpy()
    d_n = PairwiseDistance(2).forward(a, n).cpu().numpy()
    np.testing.assert_array_equal(sel.indices.cpu().numpy(), np.where(d_n - d_p < 0.1)[0])
    c1 = torch.randint(0, 64, (256,), generator=g).cuda()
    c2 = (c1 + 1 + torch.randint(0, 63, (256,), generator=g).cuda()) % 64
    labels = torch.cat([c1, c1, c2])
    idx, dist = mine_semihard_negatives(a, p, c1, e1, labels)
    idx_h, lab_h, c1_h = idx.cpu().numpy(), labels.cpu().numpy(), c1.cpu().numpy()
    assert (idx_h >= 0).all() and (lab_h[idx_h] != c1_h).all()
    d_all = torch.cdist(a.double(), e1.double()).cpu().numpy()
    other = lab_h[None, :] != c1_h[:, None]
    semi = other & (d_all > d_p[:, None].astype(np.float64) + 1e-4)
    has_semi = semi.any(axis=1)
    assert (dist.cpu().numpy()[has_semi] > d_p[has_semi]).all()
    best = np.where(semi, d_all, np.inf).min(axis=1)
    # closest among the semi-hard ones (candidates within 1e-4 of d_p are semi-hard for the kernel, not for `semi`)
    assert (dist.cpu().numpy()[has_semi] <= best[has_semi] + 1e-3).all()


CONV_CASES = [
    (2, 8, 64, 9, 32, 3, 1), (1, 16, 64, 8, 16, 3, 1), (3, 8, 128, 20, 8, 3, 1), (5, 8, 128, 10, 4, 3, 1),
    (2, 8, 64, 16, 32, 5, 2), (2, 8, 128, 13, 16, 5, 2), (3, 16, 128, 7, 8, 5, 2), (70, 24, 64, 1, 1, 1, 1),
    (4, 64, 64, 80, 32, 3, 1), (4, 256, 512, 20, 8, 5, 2),
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_kernel_vs_oracle(dev, case):
    from deepspeaker_pytorch_amd.model import get_engine
    eng = get_engine()
    b, ci, co, h, w, k, s = case
    rs = np.random.RandomState(abs(hash(case)) % 2**31)
    x = rs.randn(b, ci, h, w).astype(np.float32)
    wt = (rs.randn(co, ci, k, k) / np.sqrt(ci * k * k)).astype(np.float32)
    tw = torch.from_numpy(wt).cuda()
    wp = torch.empty(wt.size, device="cuda")
    eng.lib.call("ds_pack_conv_weight_f32", eng._p(tw), eng._p(wp), co, ci, k, 0, eng._stream(tw))
    xh = torch.from_numpy(np.ascontiguousarray(x.transpose(0, 2, 3, 1))).cuda()
    y, stats = eng.conv(xh, wp, b, h, w, ci, co, k, s, want_stats=True)
    ref = O.conv2d(x.astype(np.float64), wt.astype(np.float64), s, k // 2)
    assert rel_err(y.cpu().numpy().transpose(0, 3, 1, 2), ref) < 1e-5
    tot = stats.double().sum(dim=0).cpu().numpy()
    np.testing.assert_allclose(tot[:, 0], ref.sum(axis=(0, 2, 3)), rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(tot[:, 1], (ref * ref).sum(axis=(0, 2, 3)), rtol=1e-4, atol=1e-3)


def test_layout_roundtrip(dev):
    from deepspeaker_pytorch_amd.model import get_engine
    eng = get_engine()
    x = torch.randn(3, 5, 7, 4, device="cuda")
    y = torch.empty(3, 7, 4, 5, device="cuda")
    z = torch.empty_like(x)
    eng.lib.call("ds_nchw_to_nhwc_f32", eng._p(x), eng._p(y), 3, 5, 7, 4, eng._stream(x))
    eng.lib.call("ds_nhwc_to_nchw_f32", eng._p(y), eng._p(z), 3, 5, 7, 4, eng._stream(x))
    assert torch.equal(y, x.permute(0, 2, 3, 1).contiguous()) and torch.equal(z, x)


def grad_digest(a):
    a = np.asarray(a, np.float64).ravel()
    stride = max(1, a.size // 64)
    return np.concatenate([[np.sqrt((a * a).sum()), a.sum()], a[:16], a[::stride][:64]])


def test_backward_single_forward(dev, golden):
    """Network backward from a fixed embedding gradient (B = 8, T = 160) vs (i) the numpy oracle in float64
    on the same inputs and (ii) the digests recorded from the reference evaluated in float64 / float32.
    A clipped-ReLU mask that flips between an fp32 and an fp64 forward moves early-layer gradients by up
    to ~1e-2 (the reference's own fp32 run is that far from its fp64 run, tests/test_oracle_golden.py),
    so the late layers are held tight and the early ones to that noise band."""
    sd = O.make_state_dict(seed=31, num_classes=16)
    m = build_model(sd).train()
    x = O.make_input(seed=32, batch=8)
    ge = np.random.RandomState(77).randn(8, 512).astype(np.float32)
    e = m(torch.from_numpy(x).cuda())
    assert rel_err(e.detach().cpu().numpy(), golden["single_train_emb"]) < 5e-5
    m.zero_grad()
    e.backward(torch.from_numpy(ge).cuda())
    cache = {}
    O.forward(sd, x, train=True, dtype=np.float64, cache=cache)
    ref = O.backward(sd, cache, x, ge)
    worst = {}
    for name, p in m.named_parameters():
        if name.startswith("model.classifier"):
            assert p.grad is None
            continue
        got = p.grad.cpu().numpy()
        assert got.shape == ref[name].shape
        worst[name] = np.linalg.norm(got - ref[name]) / np.linalg.norm(ref[name])
        d64, d32 = golden["single_train64_grad/" + name], golden["single_train_grad/" + name]
        dg = grad_digest(got)
        assert np.abs(dg - d64).max() <= 8e-2 * np.abs(d64).max(), name
        assert np.abs(dg - d32).max() <= 8e-2 * np.abs(d32).max(), name
    print({k: f"{v:.1e}" for k, v in worst.items()})
    # upstream of the first clipped-ReLU mask the kernels must be tight; below it, ONE mask at the stage-4
    # output flips between the fp32 and the fp64 forward on this fixture (measured on the reference itself:
    # its fp32 and fp64 runs disagree on exactly one of 163840 masks), which moves everything below by ~5e-3
    for name in ("model.fc.weight", "model.fc.bias", "model.layer4.0.bn2.weight"):
        assert worst[name] < 1e-4, (name, worst[name])
    assert max(worst.values()) < 3e-2


def test_training_step_like_the_reference_loop(dev, golden):
    """train_triplet.py:215-224: three train-mode forwards, TripletMarginLoss, backward, Adagrad step."""
    from deepspeaker_pytorch_amd.model import TripletMarginLoss
    sd = O.make_state_dict(seed=31, num_classes=16)
    m = build_model(sd).train()
    opt = torch.optim.Adagrad(m.parameters(), lr=0.1, lr_decay=1e-4)          # train_triplet.py:372-375
    xs = [torch.from_numpy(O.make_input(seed=32 + i, batch=8)).cuda() for i in range(3)]
    out_a, out_p, out_n = m(xs[0]), m(xs[1]), m(xs[2])
    loss = TripletMarginLoss(0.1).forward(out_a, out_p, out_n)
    ref_loss = float(golden["full_train_loss"])
    assert abs(float(loss.detach()) - ref_loss) <= 1e-3 * abs(ref_loss)
    opt.zero_grad()
    loss.backward()
    n_checked = 0
    for name, p in m.named_parameters():
        if name.startswith("model.classifier"):
            continue
        ref = golden["full_train_grad/" + name]
        dg = grad_digest(p.grad.cpu().numpy())
        assert np.abs(dg - ref).max() <= 8e-2 * np.abs(ref).max(), name
        n_checked += 1
    assert n_checked == 38
    before = m.model.conv4.weight.detach().clone()
    opt.step()
    assert not torch.equal(before, m.model.conv4.weight.detach())
    assert int(m.model.bn1.num_batches_tracked) == 3
    # the next forward sees the updated weights (packed-filter cache is keyed on parameter versions)
    e2 = m(xs[0])
    assert not torch.allclose(e2.detach(), out_a.detach())
    # checkpoint round trip (train_triplet.py:325-327, 177-186)
    ck = {"epoch": 1, "state_dict": m.state_dict(), "optimizer": opt.state_dict()}
    m2 = build_model(sd)
    m2.load_state_dict(ck["state_dict"])
    m.eval(), m2.eval()
    with torch.no_grad():
        assert torch.equal(m(xs[0]), m2(xs[0]))


BWD_KERNEL_CASES = [(4, 64, 64, 80, 32, 3, 1), (4, 128, 256, 40, 16, 5, 2), (6, 512, 512, 10, 4, 3, 1),
                    (3, 64, 128, 25, 16, 5, 2)]


@pytest.mark.parametrize("case", BWD_KERNEL_CASES)
def test_conv_backward_kernels_vs_oracle(dev, case):
    from deepspeaker_pytorch_amd._native import ConvShape
    from deepspeaker_pytorch_amd.backward import _dgrad, _wgrad
    from deepspeaker_pytorch_amd.model import get_engine
    eng = get_engine()
    b, ci, co, h, w, k, s = case
    rs = np.random.RandomState(abs(hash(case)) % 2**31)
    x = rs.randn(b, ci, h, w)
    wt = rs.randn(co, ci, k, k) / np.sqrt(ci * k * k)
    ho, wo = O.conv_out_size(h, k, s, k // 2), O.conv_out_size(w, k, s, k // 2)
    gy = rs.randn(b, co, ho, wo)
    gx_ref, gw_ref = O.conv2d_bwd(x, wt, gy, s, k // 2)
    shp = ConvShape(b, h, w, ci, co, k, s)
    tw = torch.from_numpy(wt.astype(np.float32)).cuda()
    wp = torch.empty(wt.size, device="cuda")
    if s == 1:
        eng.lib.call("ds_pack_conv_weight_f32", eng._p(tw), eng._p(wp), co, ci, k, 1, eng._stream(tw))
    else:
        eng.lib.call("ds_pack_conv_dgrad_s2_f32", eng._p(tw), eng._p(wp), co, ci, eng._stream(tw))
    xh = torch.from_numpy(np.ascontiguousarray(x.transpose(0, 2, 3, 1)).astype(np.float32)).cuda()
    gyh = torch.from_numpy(np.ascontiguousarray(gy.transpose(0, 2, 3, 1)).astype(np.float32)).cuda()
    gx = _dgrad(eng, shp, gyh, wp)
    gw = _wgrad(eng, shp, xh, gyh, (co, ci, k, k))
    assert rel_err(gx.cpu().numpy().transpose(0, 3, 1, 2), gx_ref) < 1e-5
    assert rel_err(gw.cpu().numpy(), gw_ref) < 1e-5
    # split-operand bf16 matrix cores
    n = wt.size if s == 1 else 36 * co * ci
    whi = torch.empty(n, dtype=torch.bfloat16, device="cuda")
    wlo = torch.empty_like(whi)
    if s == 1:
        eng.lib.call("ds_pack_conv_weight_dgrad_bf16", eng._p(tw), eng._p(whi), eng._p(wlo), co, ci, k, eng._stream(tw))
    else:
        eng.lib.call("ds_pack_conv_weight_dgrad_s2_bf16", eng._p(tw), eng._p(whi), eng._p(wlo), co, ci, eng._stream(tw))
    gx3 = _dgrad(eng, shp, gyh, None, (whi, wlo))
    assert rel_err(gx3.cpu().numpy().transpose(0, 3, 1, 2), gx_ref) < 3e-5
    gw3 = _wgrad(eng, shp, xh, gyh, (co, ci, k, k), x3=True)
    assert rel_err(gw3.cpu().numpy(), gw_ref) < 3e-5
    assert torch.equal(gw3, _wgrad(eng, shp, xh, gyh, (co, ci, k, k), x3=True))     # deterministic


# (anchors, candidates): 8 / 4 / 2 anchors per workgroup (the launch picks what fills the chip); 256 x 6144 is the
# cross-GPU search of BASELINE configs[2]: 256 local anchors against the all-gathered embeddings of 8 ranks
@pytest.mark.parametrize("N,M", [(64, 1536), (256, 768), (256, 6144), (300, 70000 // 64), (5, 257)])
def test_mining_kernel_vs_oracle(dev, N, M):
    from deepspeaker_pytorch_amd.mining import mine_semihard_negatives
    rs = np.random.RandomState(9 + N + M)
    a = rs.randn(N, 512).astype(np.float32)
    p = a + rs.randn(N, 512).astype(np.float32) * 0.9
    cand = rs.randn(M, 512).astype(np.float32)
    la, lc = rs.randint(0, 8, N).astype(np.int64), rs.randint(0, 8, M).astype(np.int64)
    idx, dd = mine_semihard_negatives(torch.from_numpy(a).cuda(), torch.from_numpy(p).cuda(), torch.from_numpy(la).cuda(),
                                      torch.from_numpy(cand).cuda(), torch.from_numpy(lc).cuda())
    d_p = O.pairwise_distance(a, p)
    ref = O.mine_semihard(a, d_p, la, cand, lc)
    np.testing.assert_array_equal(idx.cpu().numpy(), ref)
    ref_d = np.sqrt(((a - cand[ref]) ** 2).sum(1) + 1e-4 / 512)
    assert rel_err(dd.cpu().numpy(), ref_d) < 1e-5
    # enqueued on the side stream (the caller's stream goes on; consumers are ordered after the search): same result
    mined = mine_semihard_negatives(torch.from_numpy(a).cuda(), torch.from_numpy(p).cuda(), torch.from_numpy(la).cuda(),
                                    torch.from_numpy(cand).cuda(), torch.from_numpy(lc).cuda(), side_stream=True)
    idx2, dd2 = mined
    assert torch.equal(idx2, idx) and torch.equal(dd2, dd) and torch.equal(mined.indices, idx)


@pytest.mark.parametrize("streams", [False, True])
@pytest.mark.parametrize("precision", ["f32", "bf16x3"])
def test_data_parallel_world1_nccl(dev, precision, streams):
    """Every data-parallel branch on RCCL with a single rank (Reducer(force=True)): float64 sums -> all-reduce ->
    *_from_sums kernels in the forward, the grouped BatchNorm backward split at its all-reduce, the gradient buckets
    reduced from inside the backward pass on the filter-gradient stream.  The step must equal the plain step and issue
    exactly the collectives an N-rank job issues."""
    import torch.distributed as dist
    from deepspeaker_pytorch_amd.model import TripletMarginLoss, get_engine
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29517")
    if not dist.is_initialized():
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    sd = O.make_state_dict(seed=31, num_classes=16)
    xs = [torch.from_numpy(O.make_input(seed=32 + i, batch=4)).cuda() for i in range(3)]
    lib = get_engine().lib
    from deepspeaker_pytorch_amd.engine import Engine
    for grouped in (False, True):
        grads, losses, stats = [], [], []
        for dp in (False, True):
            # both steps of a pair take the same forward -- lock-step over one batch, or one stream per member (under
            # data parallelism the streams meet at each layer's all-reduce): the two forms may tile a layer differently,
            # the statistics then differ in the last bit, and one flipped clip mask is 4e-3 on conv1's gradient
            Engine.MEMBER_STREAMS = streams
            m = build_model(sd).train()
            m.precision = precision
            red = m.enable_data_parallel(force=True) if dp else None
            assert red is None or (red.active and red.world == 1)
            lib.trace = {}
            outs = m.forward_triplet(*xs) if grouped else (m(xs[0]), m(xs[1]), m(xs[2]))
            loss = TripletMarginLoss(0.1).forward(*outs)
            loss.backward()
            trace, lib.trace = lib.trace, None
            if dp:
                m.allreduce_gradients()
                # 12 BatchNorm layers x (forward + backward) x (1 grouped | 3 separate forwards) + 5 gradient buckets
                # (per backward pass: 1 grouped | 3)
                k = 1 if grouped else 3
                assert red.n_all_reduce == k * (2 * 12 + 5), red.n_all_reduce
                # bf16x3: the reductions of 8 of the 12 layers run inside the data-gradient kernel above them
                # (ds_conv_dgrad_bnbwd_bf16) wherever its tiles do not straddle members
                # (and of 3 more inside the 5x5 stride-2 data gradient of the stage above: ds_conv_dgrad_s2_bnbwd_bf16)
                fused = trace.get("ds_conv_dgrad_bnbwd_bf16", 0) + trace.get("ds_conv_dgrad_s2_bnbwd_bf16", 0)
                assert fused == 0 if precision == "f32" else 4 * k <= fused <= 11 * k, fused
                if grouped:     # the grouped launch sequence, split at the all-reduce -- no per-member fallback
                    assert trace.get("ds_bn_bwd_group_reduce_f32", 0) + fused == 12 and trace.get("ds_bn_bwd_group_apply_f32") == 12
                    assert "ds_bn_bwd_reduce_f32" not in trace
                    # forward: one fold per layer for all members (lock-step) or one per layer and member (streams)
                    assert (trace.get("ds_partial_sum_f64_group", 0) + trace.get("ds_partial_sum_f64", 0) // 3
                            == (36 if streams else 12) + fused)
                else:
                    assert trace.get("ds_bn_bwd_reduce_f32", 0) + fused == 36 and trace.get("ds_bn_stats_from_sums_f32") == 36
            Engine.MEMBER_STREAMS = True
            grads.append({n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None})
            losses.append(float(loss))
            stats.append({k: v.clone() for k, v in m.state_dict().items() if "running" in k})
        assert abs(losses[0] - losses[1]) <= 1e-5 * max(1.0, abs(losses[0]))
        for n in grads[0]:
            assert rel_err(grads[1][n].cpu().numpy(), grads[0][n].cpu().numpy()) < 2e-5, (grouped, n)
        for k in stats[0]:
            assert rel_err(stats[1][k].cpu().numpy(), stats[0][k].cpu().numpy()) < 1e-6, (grouped, k)
    # the differentiable all-gather of the mining step and its adjoint, on RCCL's own kernels
    # (all_gather_into_tensor / reduce_scatter_tensor; gloo in the CPU tests takes a fall-back for the latter)
    from deepspeaker_pytorch_amd.distributed import AllGatherRows, Reducer
    r = Reducer()
    t = torch.randn(6, 512, device="cuda", requires_grad=True)
    gathered = AllGatherRows.apply(t, r)
    assert torch.equal(gathered, t.detach())
    gathered.backward(torch.full_like(gathered, 2.0))
    assert torch.equal(t.grad, torch.full_like(t, 2.0))
    assert torch.equal(r.reduce_scatter_rows(t.detach()), t.detach())
    dist.destroy_process_group()


@pytest.mark.parametrize("precision", ["f32", "bf16x3"])
def test_forward_triplet_equals_three_calls(dev, precision):
    """model.forward_triplet (one concatenated batch, three BatchNorm statistic sets, one backward pass) against
    the reference's call pattern model(a), model(p), model(n) + accumulated backward (train_triplet.py:215-224)."""
    from deepspeaker_pytorch_amd.model import DeepSpeakerModel, TripletMarginLoss
    sd = O.make_state_dict(seed=31, num_classes=16)
    xs = [torch.from_numpy(O.make_input(seed=32 + i, batch=6)).cuda() for i in range(3)]
    res = []
    for grouped in (False, True):
        m = DeepSpeakerModel(512, 16, precision=precision)
        m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
        m = m.cuda().train()
        outs = m.forward_triplet(*xs) if grouped else (m(xs[0]), m(xs[1]), m(xs[2]))
        loss = TripletMarginLoss(0.1).forward(*outs)
        loss.backward()
        res.append((outs, loss, {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None},
                    {k: v.clone() for k, v in m.state_dict().items() if "running" in k or "num_batches" in k}))
    (o0, l0, g0, s0), (o1, l1, g1, s1) = res
    # per pixel the convolutions are the same arithmetic; the batch statistics are folded from per-tile partial
    # sums, and the tiling of an 18-utterance launch differs from that of a 6-utterance one: last-bit differences
    for a, b in zip(o0, o1):
        assert rel_err(b.detach().cpu().numpy(), a.detach().cpu().numpy()) < 2e-5
    assert abs(float(l0) - float(l1)) < 2e-5 * max(1.0, abs(float(l0)))
    for k in s0:                                                   # three sequential running-statistics updates
        assert rel_err(s1[k].double().cpu().numpy(), s0[k].double().cpu().numpy()) < 1e-6, k
    assert int(s1["model.bn1.num_batches_tracked"]) == 3
    for n in g0:
        err = float((g1[n] - g0[n]).norm() / g0[n].norm().clamp_min(1e-30))
        assert err < 1e-2, (n, err)         # one contraction over all pixels vs three summed; a clip mask that flips
                                            # on a last-bit statistics difference moves early layers by ~5e-3


@pytest.mark.parametrize("precision,tol", [("bf16x3", 1e-4), ("bf16", 3e-2)])
def test_bf16_precisions_vs_reference(dev, golden, precision, tol):
    """The bf16 matrix-core variants of the eval forward: bf16x3 must stay inside the 1e-3 contract with
    margin (it is f32-class), plain bf16 is a speed mode with its own stated tolerance."""
    from deepspeaker_pytorch_amd.model import DeepSpeakerModel
    sd = O.make_state_dict(seed=11, num_classes=16)
    m = DeepSpeakerModel(512, 16, precision=precision)
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    m = m.cuda().eval()
    x = O.make_input(seed=12, batch=6)
    with torch.no_grad():
        e = m(torch.from_numpy(x).cuda()).cpu().numpy()
    err = rel_err(e, golden["full_eval_emb"])
    print(precision, "embedding rel err vs reference:", err)
    assert err < tol
    for T in (100, 237):
        xv = O.make_input(seed=100 + T, batch=2, frames=T)
        with torch.no_grad():
            ev = m(torch.from_numpy(xv).cuda()).cpu().numpy()
        assert rel_err(ev, golden[f"full_eval_T{T}_emb"]) < tol
    if precision == "bf16x3":            # selection identity on the loss-side fixture embeddings still holds
        from deepspeaker_pytorch_amd.mining import select_triplets
        sdt = O.make_state_dict(seed=31, num_classes=16)
        mt = DeepSpeakerModel(512, 16, precision=precision)
        mt.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sdt.items()})
        mt = mt.cuda().eval()
        xs = [torch.from_numpy(O.make_input(seed=32 + i, batch=8)).cuda() for i in range(3)]
        with torch.no_grad():
            embs = [mt(t) for t in xs]
        refs = [O.forward(sdt, t.cpu().numpy(), dtype=np.float64).astype(np.float32) for t in xs]
        _, d_p, d_n = O.triplet_margin_loss(*refs, margin=0.1)
        ref_idx, _, _ = O.triplet_filter(d_p, d_n, 0.1)
        sel = select_triplets(*embs, margin=0.1)
        np.testing.assert_array_equal(sel.indices.cpu().numpy(), ref_idx)


def test_training_step_bf16x3(dev, golden):
    """bf16x3 training (forward, data and filter gradients on the bf16 matrix cores): loss and gradients stay within
    the fp32 noise band of the reference (same bounds as the f32 training test)."""
    from deepspeaker_pytorch_amd.model import DeepSpeakerModel, TripletMarginLoss
    sd = O.make_state_dict(seed=31, num_classes=16)
    m = DeepSpeakerModel(512, 16, precision="bf16x3")
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    m = m.cuda().train()
    xs = [torch.from_numpy(O.make_input(seed=32 + i, batch=8)).cuda() for i in range(3)]
    outs = [m(x) for x in xs]
    for e, k in zip(outs, "apn"):
        assert rel_err(e.detach().cpu().numpy(), golden[f"full_train_emb_{k}"]) < 1e-4
    loss = TripletMarginLoss(0.1).forward(*outs)
    ref_loss = float(golden["full_train_loss"])
    assert abs(float(loss.detach()) - ref_loss) <= 1e-3 * abs(ref_loss)
    loss.backward()
    for name, p in m.named_parameters():
        if name.startswith("model.classifier"):
            continue
        ref = golden["full_train_grad/" + name]
        dg = grad_digest(p.grad.cpu().numpy())
        # Clipped-ReLU masks that flip under a 1e-5 perturbation of the activations move individual gradient
        # entries by a few percent of the tensor's largest (the reference's own fp32 run is 3.5 % from its
        # fp64 run on layer3.0.conv1.weight); the tensor NORM is insensitive to those sparse flips and is
        # held tight: measured 1e-5 .. 1.4e-3 over all tensors, against the exact-f32 path's 7e-6 .. 9e-4.
        assert abs(dg[0] - ref[0]) <= 5e-3 * abs(ref[0]), name
        assert np.abs(dg - ref).max() <= 1.5e-1 * np.abs(ref).max(), name


@pytest.mark.parametrize("case", [(768, 512, 512, 10, 4, 3, 1), (768, 256, 512, 20, 8, 5, 2), (96, 512, 1024, 10, 4, 3, 1)])
def test_xcd_tile_queues_and_wide_register_tile_are_bitwise_the_one_tile_kernel(case):
    """Round 5 at the bench size on the real chip: the persistent fp16 convolution with one tile queue per
    XCD and, for the 512-channel 3x3 layers, the 128-channel-wide register tile (NSUB = 4) -- bit-identical to its own
    one-queue / 64-wide forms and to the one-tile-per-workgroup kernel, and within 1e-5 of a float64 convolution of the
    same fp16 operands on a sample of images.  (The third case is the 5x5 stride-2 data gradient of the fp16 training
    step run as a 3x3 convolution with 4 Cin output channels.)"""
    import ctypes
    from deepspeaker_pytorch_amd._native import (ConvShape, DS_CONV_HINT_NO_PERSIST, DS_CONV_HINT_NO_WIDE, DS_CONV_HINT_ONE_QUEUE,
                                                 DS_EPI_AFFINE, DS_EPI_CLIP)
    from deepspeaker_pytorch_amd.model import get_engine
    eng = get_engine()
    b, ci, co, h, w, k, s_ = case
    ho, wo = (h - 1) // s_ + 1, (w - 1) // s_ + 1
    g = torch.Generator(device="cpu").manual_seed(sum(case))
    x = torch.randn(b, h, w, ci, generator=g).abs().half().cuda()
    wt = (torch.randn(co, ci, k, k, generator=g) * (1.0 / (ci * k * k) ** 0.5)).cuda()
    wp = eng._pack_f16(wt, k)
    sc, sh = (torch.rand(co, generator=g) + 0.5).cuda(), torch.randn(co, generator=g).cuda()
    shp = ConvShape(b, h, w, ci, co, k, s_)
    st = eng._stream(x)
    outs = {}
    for name, hint in (("default", 0), ("one queue", DS_CONV_HINT_ONE_QUEUE), ("64-wide", DS_CONV_HINT_NO_WIDE),
                       ("one tile per workgroup", DS_CONV_HINT_NO_PERSIST)):
        y = torch.full((b, ho, wo, co), float("nan"), dtype=torch.float16, device="cuda")
        eng.lib.call("ds_conv_fwd_f16", ctypes.byref(shp), eng._p(x), eng._p(wp), eng._p(sc), eng._p(sh), None, eng._p(y),
                     DS_EPI_AFFINE | DS_EPI_CLIP | hint, st)
        outs[name] = y
    torch.cuda.synchronize()
    for name, y in outs.items():
        assert bool(torch.isfinite(y.float()).all()), name
        assert torch.equal(y, outs["default"]), name
    out8 = (ctypes.c_int * 8)()
    eng.lib.call("ds_conv_f16_plan_describe", ctypes.byref(shp), out8)
    # (a two-wave workgroup on a 256-channel n tile is the NSUB = 4 plan; the 5x5 layer's 128 x 256 plan has four waves)
    assert out8[7] >= 10000 and ((out8[1], out8[6]) == (256, 128)) == (k == 3 and co % 256 == 0), list(out8)
    nb = 4
    ref = torch.nn.functional.conv2d(x[:nb].permute(0, 3, 1, 2).double().cpu(), wt.half().double().cpu(), None, s_, k // 2)
    ref = (ref * sc.double().cpu()[None, :, None, None] + sh.double().cpu()[None, :, None, None]).clamp(0, 20)
    got = outs["default"][:nb].permute(0, 3, 1, 2).double().cpu()
    assert float((got - ref).abs().max()) <= 20 * 2.0 ** -11 + 1e-5
