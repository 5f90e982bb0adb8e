"""BASELINE.json configs[1] on a real MI355X against outputs of the UNMODIFIED reference at that configuration
(tests/golden/reference_cfg1.npz, made by tests/golden/make_golden.py): all 768 embeddings, the triplet loss and
the filter's selection over the 256 triplets, for every arithmetic the eval forward offers.  Contract
(north_star): embeddings and loss within 1e-3 relative, identical selection."""
import ctypes
import os

import numpy as np
import pytest
import torch

import deepspeaker_oracle as O
from conftest import ROOT, rel_err

pytestmark = pytest.mark.gpu

CONTRACT = 1e-3
# measured bars per arithmetic (embeddings, max |diff| / max |ref| over all 768 rows)
EMB_BAR = {"f32": 2e-5, "bf16x3": 3e-5, "f16": CONTRACT}


@pytest.fixture(scope="module")
def cfg1():
    g = np.load(os.path.join(ROOT, "tests", "golden", "reference_cfg1.npz"))
    gen = torch.Generator(device="cpu").manual_seed(1234)
    x = torch.randn(768, 1, 160, 64, generator=gen)
    dig = g["cfg1_input_digest"]
    # the inputs are regenerated, not stored: make sure this torch build draws the same stream
    assert abs(float(x.double().sum()) - dig[0]) < 1e-6 * max(1.0, abs(dig[0])) and float(x[767, 0, 159, 63]) == dig[2]
    return g, O.make_state_dict(seed=0, num_classes=1211), x.cuda()


def build(sd, precision):
    from deepspeaker_pytorch_amd.model import DeepSpeakerModel
    m = DeepSpeakerModel(512, 1211, precision=precision)
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    return m.cuda().eval()


@pytest.mark.parametrize("precision", ["f32", "bf16x3", "f16"])
def test_bench_size_vs_reference_golden(cfg1, precision):
    from deepspeaker_pytorch_amd.mining import select_triplets
    from deepspeaker_pytorch_amd.model import TripletMarginLoss
    g, sd, x = cfg1
    m = build(sd, precision)
    with torch.no_grad():
        e = m(x).clone()
        a, p, n = e[:256], e[256:512], e[512:]
        loss = TripletMarginLoss(0.1).forward(a, p, n)
        sel = select_triplets(a, p, n, 0.1, model=m, inputs=(x[:256], x[256:512], x[512:]))
    ref = g["cfg1_emb"]
    err = rel_err(e.cpu().numpy(), ref)
    row = float(((e.cpu().double() - torch.from_numpy(ref).double()).norm(dim=1) / torch.from_numpy(ref).double().norm(dim=1)).max())
    gap_ref = g["cfg1_d_n"] - g["cfg1_d_p"] - np.float32(0.1)
    gap = sel.d_n.cpu().numpy() - sel.d_p.cpu().numpy() - np.float32(0.1)
    loss_rel = abs(float(loss) - float(g["cfg1_loss"])) / float(g["cfg1_loss"])
    sel_loss_rel = abs(float(sel.loss) - float(g["cfg1_loss"])) / float(g["cfg1_loss"])
    print(f"\\n[{precision}] embeddings: max|d|/max {err:.3e}, worst row rel-L2 {row:.3e}; loss rel {loss_rel:.3e} "
          f"(after refinement {sel_loss_rel:.3e}); reference min |d_n-d_p-margin| {np.abs(gap_ref).min():.3e}; "
          f"max |gap error| {np.abs(gap - gap_ref).max():.3e}; near ties refined: "
          f"{int(sel.amb_count) if sel.amb_count is not None else 'n/a'}")
    assert err < EMB_BAR[precision] and row < EMB_BAR[precision]
    assert loss_rel < CONTRACT and sel_loss_rel < CONTRACT
    np.testing.assert_array_equal(sel.indices.cpu().numpy(), g["cfg1_selected"])        # identical selection
    assert abs(float(sel.mean_diff) - float(g["cfg1_mean_diff"])) < 1e-3 * abs(float(g["cfg1_mean_diff"])) + 1e-5
    if precision == "f16":
        assert not sel.refine_overflow
        # the refinement band must cover what fp16 does to the decision variable
        d_p16, d_n16 = (select_triplets(a, p, n, 0.1).d_p.cpu().numpy(), select_triplets(a, p, n, 0.1).d_n.cpu().numpy())
        from deepspeaker_pytorch_amd.mining import REFINE_BAND
        assert np.abs(d_n16 - d_p16 - np.float32(0.1) - gap_ref).max() < 0.75 * REFINE_BAND


def test_f16_refinement_decides_planted_near_ties(cfg1):
    """Force near ties: choose the margin so that triplets sit exactly on the fp16 path's decision boundary; the
    refined selection must be the selection of the f32-class distances for every triplet inside the band."""
    from deepspeaker_pytorch_amd.mining import REFINE_BAND, select_triplets
    g, sd, x = cfg1
    m16, m3 = build(sd, "f16"), build(sd, "bf16x3")
    with torch.no_grad():
        e16, e3 = m16(x).clone(), m3(x).clone()
    d3 = select_triplets(e3[:256], e3[256:512], e3[512:], 0.1)
    diff3 = (d3.d_n - d3.d_p).cpu().numpy()
    order = np.argsort(diff3)
    flips_seen = 0
    for k in (40, 100, 128, 200):                   # a margin half-way between two neighbouring triplets
        margin = float((diff3[order[k]] + diff3[order[k + 1]]) / 2)
        with torch.no_grad():
            plain = select_triplets(e16[:256], e16[256:512], e16[512:], margin)
            fine = select_triplets(e16[:256], e16[256:512], e16[512:], margin, model=m16,
                                   inputs=(x[:256], x[256:512], x[512:]), cap=32)
        want = np.where(diff3 < np.float32(margin))[0]
        got = fine.indices.cpu().numpy()
        assert not fine.refine_overflow
        np.testing.assert_array_equal(got, want)
        flips_seen += int(len(plain.indices) != len(want) or (plain.indices.cpu().numpy() != want).any())
        assert int(fine.amb_count) >= 2              # the two triplets the margin was placed between
    print("\\nunrefined fp16 selections that differed from the f32-class one:", flips_seen, "of 4")


def _planted(x, m3, n_plant, target):
    """The cfg1 batch with the negatives of the first n_plant triplets replaced by slightly perturbed copies of their
    positives, the perturbation scaled so that median |d_n - d_p| of those triplets is `target` (f32-class forward):
    d_n - d_p concentrated at one value, as a triplet-trained network concentrates it at the margin."""
    from deepspeaker_pytorch_amd.mining import select_triplets
    xa, xp, xn = x[:256].clone(), x[256:512].clone(), x[512:].clone()
    noise = torch.randn(n_plant, 1, 160, 64, generator=torch.Generator(device="cpu").manual_seed(77)).cuda()
    delta = 1e-3
    for _ in range(4):
        xn[:n_plant] = xp[:n_plant] + delta * noise
        with torch.no_grad():
            e = m3(torch.cat([xa, xp, xn])).clone()
        d = select_triplets(e[:256], e[256:512], e[512:], 0.0)
        diff = (d.d_n - d.d_p).cpu().numpy()
        med = float(np.median(np.abs(diff[:n_plant] - np.median(diff[:n_plant]))))
        if 0.8 * target < med < 1.25 * target:
            break
        delta *= target / max(med, 1e-12)
    return (xa, xp, xn), diff, float(np.median(diff[:n_plant]))


def test_f16_refinement_default_slots_many_near_ties(cfg1):
    """the DEFAULT slot count with >= 16 planted near ties -- refined selection == f32-class
    selection, no overflow."""
    from deepspeaker_pytorch_amd.mining import REFINE_CAP_START, select_triplets
    g, sd, x = cfg1
    m16, m3 = build(sd, "f16"), build(sd, "bf16x3")
    (xa, xp, xn), diff3, margin = _planted(x, m3, 24, 0.6e-3)
    with torch.no_grad():
        e16 = m16(torch.cat([xa, xp, xn])).clone()
        fine = select_triplets(e16[:256], e16[256:512], e16[512:], margin, model=m16, inputs=(xa, xp, xn))
    want = np.where(diff3 < np.float32(margin))[0]
    print(f"\nplanted 24: near ties found {fine.n_near_ties}, slots {fine.amb_cap}, selected {len(want)} of 256")
    assert fine.amb_cap == REFINE_CAP_START and 16 <= fine.n_near_ties <= REFINE_CAP_START
    assert not fine.refine_overflow and not fine.refined_all
    np.testing.assert_array_equal(fine.indices.cpu().numpy(), want)


def test_f16_refinement_margin_concentrated_batch(cfg1):
    """d_n - d_p concentrated AT the margin for most of the batch (what triplet training produces).
    The first call finds far more near ties than it has slots for: the selection then re-embeds the whole batch at
    f32-class precision before anything is read (`refined_all`), and the policy sizes the next calls from what it saw
    -- those decide every near tie in their own slots, no overflow.  Every selection equals the f32-class one."""
    from deepspeaker_pytorch_amd.mining import REFINE_CAP_START, refine_policy, select_triplets
    g, sd, x = cfg1
    m16, m3 = build(sd, "f16"), build(sd, "bf16x3")
    (xa, xp, xn), diff3, margin = _planted(x, m3, 200, 0.6e-3)
    want = np.where(diff3 < np.float32(margin))[0]
    with torch.no_grad():
        e16 = m16(torch.cat([xa, xp, xn])).clone()
        plain = select_triplets(e16[:256], e16[256:512], e16[512:], margin)
        first = select_triplets(e16[:256], e16[256:512], e16[512:], margin, model=m16, inputs=(xa, xp, xn))
        assert first.amb_cap == REFINE_CAP_START
        np.testing.assert_array_equal(first.indices.cpu().numpy(), want)       # resolves the overflow first
        assert first.refine_overflow and first.refined_all and first.n_near_ties > 64
        second = select_triplets(e16[:256], e16[256:512], e16[512:], margin, model=m16, inputs=(xa, xp, xn))
        assert second.amb_cap >= second.n_near_ties and not second.refine_overflow and not second.refined_all
        np.testing.assert_array_equal(second.indices.cpu().numpy(), want)
    pol = refine_policy(m16)
    flips = int(len(plain.indices) != len(want) or (plain.indices.cpu().numpy() != want).any())
    print(f"\nconcentrated batch: near ties {first.n_near_ties} of 256 (slots {first.amb_cap} -> {second.amb_cap}), "
          f"policy calls {pol.calls} overflows {pol.overflows}; unrefined fp16 selection differs: {bool(flips)}")
    assert pol.overflows == 1 and abs(float(second.loss) - float(first.loss)) < 1e-6


def test_f16_refinement_window_equals_per_call_refinement(cfg1):
    """Near ties of several consecutive calls through ONE refinement forward (`RefineWindow`): every tensor a selection
    hands out is bit-identical to per-call refinement; a selection read while its window is open closes it; a change
    of slot count closes the open window first."""
    from deepspeaker_pytorch_amd.mining import refine_policy, select_triplets
    g, sd, x = cfg1
    m16, m3 = build(sd, "f16"), build(sd, "bf16x3")
    with torch.no_grad():
        e3 = m3(x).clone()
    diff3 = (select_triplets(e3[:256], e3[256:512], e3[512:], 0.1).d_n
             - select_triplets(e3[:256], e3[256:512], e3[512:], 0.1).d_p).cpu().numpy()
    order = np.argsort(diff3)
    margins = [float((diff3[order[k]] + diff3[order[k + 1]]) / 2) for k in (40, 100, 128, 200, 60)]
    batches = []
    for j in range(5):                              # five different batches: the triplets rolled by j * 7
        idx = torch.roll(torch.arange(256, device="cuda"), j * 7)
        batches.append(tuple(x[o:o + 256][idx].contiguous() for o in (0, 256, 512)))
    with torch.no_grad():
        embs = [m16(torch.cat(b)).clone() for b in batches]

    def run(window, caps):
        pol = refine_policy(m16)
        pol.window = window
        pol.calls = 1000                            # the probe window rotates with the call count: the same in both runs
        out = []
        with torch.no_grad():
            for j, (b, e) in enumerate(zip(batches, embs)):
                out.append(select_triplets(e[:256], e[256:512], e[512:], margins[j], model=m16, inputs=b, cap=caps[j]))
        return out

    try:
        ref = run(1, [16] * 5)
        got = run(4, [16] * 5)                      # calls 0-3 share a forward, call 4 stays open until read
        assert got[4]._window is not None and got[0]._window is None
        assert refine_policy(m16).batch is not None
        for r, s_ in zip(ref, got):
            for name in ("indices", "d_p", "d_n", "loss", "mean_diff"):
                assert torch.equal(getattr(r, name), getattr(s_, name)), name
            assert r.n_near_ties == s_.n_near_ties and not s_.refine_overflow
            assert r.observed_error == s_.observed_error
        assert got[4]._window is None and refine_policy(m16).batch is None
        for j, s_ in enumerate(got):                # ... and both equal the f32-class selection of that batch
            want = np.where(np.roll(diff3, j * 7) < np.float32(margins[j]))[0]
            np.testing.assert_array_equal(s_.indices.cpu().numpy(), want)
        mixed = run(4, [16, 16, 32, 32, 32])          # a new slot count closes the open window (calls complete in order)
        assert mixed[0]._window is None and mixed[1]._window is None and mixed[2]._window is not None
        refine_policy(m16).flush()
        assert all(s_._window is None for s_ in mixed)
        for r, s_ in zip(ref, mixed):
            assert torch.equal(r.indices, s_.indices)
    finally:
        refine_policy(m16).window = 1


def test_selection_and_loss_on_a_graphed_output(cfg1):
    """a HIP graph's static output keeps its address and its torch version counter across replays; loss and
    selection of the second batch must be the second batch's (nothing may be served from a result keyed on identity)."""
    from deepspeaker_pytorch_amd.mining import select_triplets
    from deepspeaker_pytorch_amd.model import TripletMarginLoss
    g, sd, x = cfg1
    m = build(sd, "f32")
    x1, x2 = x[:48].contiguous(), x[48:96].contiguous()
    with torch.no_grad():
        want = []
        for xb in (x1, x2):
            e = m(xb).clone()
            want.append((float(TripletMarginLoss(0.1).forward(e[:16], e[16:32], e[32:])),
                         select_triplets(e[:16], e[16:32], e[32:], 0.1).d_n.cpu().numpy().copy()))
        graphed = m.graphed(x1)
        got = []
        for xb in (x1, x2):
            e = graphed(xb)                             # the SAME tensor object both times
            a, p, n = e[:16], e[16:32], e[32:]
            got.append((float(TripletMarginLoss(0.1).forward(a, p, n)), select_triplets(a, p, n, 0.1).d_n.cpu().numpy().copy()))
    assert abs(want[0][0] - want[1][0]) > 1e-6           # the two batches do differ
    for (wl, wd), (gl, gd) in zip(want, got):
        assert abs(wl - gl) < 1e-6 and np.abs(wd - gd).max() < 1e-5


def test_f16_forward_properties(cfg1):
    g, sd, x = cfg1
    m = build(sd, "f16")
    with torch.no_grad():
        e1, e2 = m(x).clone(), m(x).clone()
        part = m(x[256:512].contiguous()).clone()
        one = m(x[5:6].contiguous()).clone()
    assert torch.isfinite(e1).all() and torch.equal(e1, e2)                  # bitwise deterministic
    assert float((e1.double().norm(dim=1) - 10).abs().max()) < 1e-4
    assert torch.equal(e1[256:512], part) and torch.equal(e1[5:6], one)     # independent of batch composition


F16_CASES = [
    # (B, Cin, Cout, H, W, KS, stride)
    (2, 64, 64, 11, 32, 3, 1), (3, 32, 128, 20, 8, 3, 1), (5, 96, 128, 10, 4, 3, 1), (2, 64, 128, 21, 16, 5, 2),
    (3, 32, 256, 9, 8, 5, 2), (1, 64, 64, 3, 5, 3, 1), (1, 64, 128, 21, 64, 5, 2), (1, 32, 64, 12, 100, 3, 1),
    (48, 64, 64, 80, 32, 3, 1), (48, 64, 128, 80, 32, 5, 2), (32, 128, 128, 40, 16, 3, 1), (32, 128, 256, 40, 16, 5, 2),
    (32, 256, 256, 20, 8, 3, 1), (33, 256, 512, 20, 8, 5, 2), (67, 512, 512, 10, 4, 3, 1), (3, 512, 512, 50, 4, 3, 1),
    (2, 32, 128, 100, 4, 3, 1),
]


@pytest.mark.parametrize("case", F16_CASES)
@pytest.mark.parametrize("single_buffer", [0, 1, 2])
def test_conv_f16_kernel(case, single_buffer):
    """ds_conv_fwd_f16 against a float64 convolution of the same fp16-rounded operands: fp16 products are exact
    in f32, so only the f32 accumulation order separates the two (1e-5 of the largest output)."""
    from deepspeaker_pytorch_amd._native import (ConvShape, DS_CONV_HINT_CHUNK16, DS_CONV_HINT_SINGLE_BUFFER,
                                                 DS_EPI_AFFINE, DS_EPI_CLIP, DS_EPI_OUT_F32, DS_EPI_RESIDUAL)
    from deepspeaker_pytorch_amd.model import get_engine
    eng = get_engine()
    b, ci, co, h, w, k, s = case
    gen = torch.Generator(device="cpu").manual_seed(sum(case))
    x = torch.randn(b, h, w, ci, generator=gen).abs().half().cuda()
    wt = (torch.randn(co, ci, k, k, generator=gen) / (ci * k * k) ** 0.5).half().float().cuda()
    wp = eng._pack_f16(wt, k)
    ho, wo = (h + 2 * (k // 2) - k) // s + 1, (w + 2 * (k // 2) - k) // s + 1
    if single_buffer == 2 and case[5] != 5:
        pytest.skip("16-channel chunks exist for the 5x5 layers only")
    hint = (0, DS_CONV_HINT_SINGLE_BUFFER, DS_CONV_HINT_CHUNK16)[single_buffer]
    shp = ConvShape(b, h, w, ci, co, k, s)
    y = torch.full((b, ho, wo, co), float("nan"), device="cuda")
    eng.lib.call("ds_conv_fwd_f16", ctypes.byref(shp), eng._p(x), eng._p(wp), None, None, None, eng._p(y),
                 DS_EPI_OUT_F32 | hint, eng._stream(x))
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), wt.double(), None, s, k // 2).permute(0, 2, 3, 1)
    assert rel_err(y.cpu().numpy(), ref.cpu().numpy()) < 1e-5
    # full epilogue, fp16 store
    scale = (torch.rand(co, generator=gen) + 0.5).cuda()
    shift = torch.randn(co, generator=gen).cuda()
    res = (torch.randn(b, ho, wo, co, generator=gen).abs() * 6).half().cuda()
    y16 = torch.full((b, ho, wo, co), float("nan"), dtype=torch.float16, device="cuda")
    eng.lib.call("ds_conv_fwd_f16", ctypes.byref(shp), eng._p(x), eng._p(wp), eng._p(scale), eng._p(shift), eng._p(res),
                 eng._p(y16), DS_EPI_AFFINE | DS_EPI_RESIDUAL | DS_EPI_CLIP | hint, eng._stream(x))
    want = (ref * scale.double() + shift.double() + res.double()).clamp(0.0, 20.0)
    assert float((y16.double() - want).abs().max()) <= 20 * 2.0 ** -11 + 1e-4
    assert float(y16.min()) >= 0.0 and float(y16.max()) <= 20.0
