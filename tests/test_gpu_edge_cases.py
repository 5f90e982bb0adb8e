"""Edge cases of the hot path on a real MI355X, against outputs of the unmodified reference
(tests/golden/reference_outputs.npz) or hand-derived exact values: shortest / longest inputs, ragged batch sizes,
empty and full filter selections, the hinge and the clipped ReLU exactly at their kinks, whole-network gradients on
a fixture free of near-boundary clip inputs, and the softmax head with gradients."""
import ctypes

import numpy as np
import pytest
import torch

import deepspeaker_oracle as O
from conftest import rel_err

pytestmark = pytest.mark.gpu

TOL = {"f32": 2e-5, "bf16x3": 3e-5, "f16": 1e-3}          # embeddings vs the reference (contract: 1e-3)


def build(sd, precision="f32", num_classes=16):
    from deepspeaker_pytorch_amd.model import DeepSpeakerModel
    m = DeepSpeakerModel(512, num_classes, precision=precision)
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    return m.cuda()


def grad_digest(a):
    a = np.asarray(a, np.float64).ravel()
    stride = max(1, a.size // 64)
    return np.concatenate([[np.sqrt((a * a).sum()), a.sum()], a[:16], a[::stride][:64]])


@pytest.mark.parametrize("precision", ["f32", "bf16x3", "f16"])
@pytest.mark.parametrize("T", [1, 800])
def test_shortest_and_longest_utterances(golden, precision, T):
    """T = 1 (one frame: every layer is all halo) and T = 800 (the top of configs[4]'s range)"""
    sd = O.make_state_dict(seed=11, num_classes=16)
    m = build(sd, precision).eval()
    x = torch.from_numpy(O.make_input(seed=100 + T, batch=2, frames=T)).cuda()
    with torch.no_grad():
        e = m(x)
    assert rel_err(e.cpu().numpy(), golden[f"full_eval_T{T}_emb"]) < TOL[precision]


@pytest.mark.parametrize("precision", ["f32", "bf16x3", "f16"])
def test_ragged_batch_sizes(golden, precision):
    """B = 257 (one more than a power of two), 3 and 1: rows are independent in eval mode, so every size must
    reproduce the reference's rows; ragged last tiles must not leak into them."""
    sd = O.make_state_dict(seed=11, num_classes=16)
    m = build(sd, precision).eval()
    x = torch.from_numpy(O.make_input(seed=300, batch=257, frames=32)).cuda()
    ref = golden["full_eval_B257_T32_emb"]
    with torch.no_grad():
        e257 = m(x).clone()
        e3 = m(x[5:8].contiguous()).clone()
        e1 = m(x[256:257].contiguous()).clone()
    assert rel_err(e257.cpu().numpy(), ref) < TOL[precision]
    assert rel_err(e3.cpu().numpy(), ref[5:8]) < TOL[precision]
    assert rel_err(e1.cpu().numpy(), ref[256:257]) < TOL[precision]


def test_filter_selects_nothing_and_everything():
    """train_triplet.py:262-264: an empty selection (the reference skips the batch) and a full one"""
    from deepspeaker_pytorch_amd.mining import select_triplets
    rs = np.random.RandomState(3)
    a = rs.randn(37, 512).astype(np.float32)
    near, far = a + 0.01 * rs.randn(37, 512).astype(np.float32), a + 3.0 * rs.randn(37, 512).astype(np.float32)
    ta, tn, tf = (torch.from_numpy(v).cuda() for v in (a, near, far))
    none = select_triplets(ta, tn, tf, margin=0.1)          # d_n - d_p is huge: every triplet already satisfies the margin
    assert none.n_selected == 0 and none.indices.numel() == 0 and none.n_correct == 37
    every = select_triplets(ta, tf, tn, margin=0.1)         # positives far, negatives near: all violate it
    assert every.n_selected == 37 and every.indices.cpu().tolist() == list(range(37))
    d_p, d_n = O.pairwise_distance(a, far), O.pairwise_distance(a, near)
    assert abs(float(every.mean_diff) - float(np.mean(d_n - d_p))) < 1e-4
    one = select_triplets(ta[:1], tf[:1], tn[:1], margin=0.1)
    assert one.n_selected == 1


@pytest.mark.parametrize("p", [1, 2, 3])
def test_pairwise_distance_any_norm_with_autograd(p):
    """reference model.py:8-18 accepts any norm (its call sites pass 2): forward and both gradients against the same
    formula in plain torch on the CPU."""
    from deepspeaker_pytorch_amd.model import PairwiseDistance
    g = torch.Generator().manual_seed(5 + p)
    x1 = torch.randn(37, 512, generator=g)
    x2 = torch.randn(37, 512, generator=g)
    w = torch.randn(37, generator=g)
    r1, r2 = x1.clone().requires_grad_(), x2.clone().requires_grad_()
    ref = torch.pow(torch.pow(torch.abs(r1 - r2), p).sum(dim=1) + 1e-4 / 512, 1.0 / p)       # model.py:15-18
    (ref * w).sum().backward()
    d1, d2 = x1.cuda().requires_grad_(), x2.cuda().requires_grad_()
    out = PairwiseDistance(p).forward(d1, d2)
    (out * w.cuda()).sum().backward()
    assert torch.allclose(out.detach().cpu(), ref.detach(), rtol=2e-6, atol=1e-6)
    assert torch.allclose(d1.grad.cpu(), r1.grad, rtol=1e-5, atol=1e-6)
    assert torch.allclose(d2.grad.cpu(), r2.grad, rtol=1e-5, atol=1e-6)


def test_hinge_exactly_zero_has_subgradient_one():
    """TripletMarginLoss at the kink (model.py:30-31, clamp(min=0)): with margin 0 and positive == negative the hinge
    argument is exactly 0; torch's clamp passes the gradient there, so d_p and d_n still get +-1/N."""
    from deepspeaker_pytorch_amd.model import TripletMarginLoss
    rs = np.random.RandomState(5)
    a = torch.from_numpy(rs.randn(4, 512).astype(np.float32)).cuda().requires_grad_(True)
    pn = rs.randn(4, 512).astype(np.float32)
    p = torch.from_numpy(pn).cuda().requires_grad_(True)
    n = torch.from_numpy(pn.copy()).cuda().requires_grad_(True)
    loss = TripletMarginLoss(0.0).forward(a, p, n)
    assert float(loss) == 0.0
    loss.backward()
    ra, rp, rn = (torch.from_numpy(v.detach().cpu().numpy()).requires_grad_(True) for v in (a, p, n))
    d = lambda x, y: torch.sqrt(((x - y).abs() ** 2).sum(1) + 1e-4 / 512)          # model.py:13-18
    torch.clamp(0.0 + d(ra, rp) - d(ra, rn), min=0.0).mean().backward()           # model.py:27-33
    assert float(rp.grad.abs().max()) > 0                                          # the kink passes gradient
    for got, ref in ((a.grad, ra.grad), (p.grad, rp.grad), (n.grad, rn.grad)):
        assert float((got.cpu() - ref).abs().max()) < 1e-6


@pytest.mark.parametrize("arith", ["f32", "bf16x3", "f16"])
def test_clip_exactly_at_zero_and_twenty_forward(arith):
    """conv epilogue outputs that land EXACTLY on 0 and 20 (model.py:36-44): a one-hot filter copies the input,
    scale / shift put chosen pixels on the boundaries and one ulp either side of them."""
    from deepspeaker_pytorch_amd._native import ConvShape, DS_EPI_AFFINE, DS_EPI_CLIP, DS_EPI_OUT_F32
    from deepspeaker_pytorch_amd.model import get_engine
    eng = get_engine()
    B, H, W, C = 1, 4, 8, 64
    # (values with short mantissas: exact in fp16 and in the two-term bf16 split alike)
    vals = np.array([0.0, 20.0, -1.0, 21.0, 0.5, 19.5, 2.0 ** -10, 20.0 - 2.0 ** -6], np.float32)
    x = np.zeros((B, H, W, C), np.float32)
    x[0, 1, :, 3] = vals
    w = np.zeros((C, C, 3, 3), np.float32)
    for c in range(C):
        w[c, c, 1, 1] = 1.0                                   # identity convolution
    scale, shift = torch.ones(C).cuda(), torch.zeros(C).cuda()
    shp = ConvShape(B, H, W, C, C, 3, 1)
    wt = torch.from_numpy(w).cuda()
    y = torch.full((B, H, W, C), float("nan")).cuda()
    st = eng._stream(y)
    if arith == "f16":
        wp = eng._pack_f16(wt, 3)
        eng.lib.call("ds_conv_fwd_f16", ctypes.byref(shp), eng._p(torch.from_numpy(x).half().cuda()), eng._p(wp),
                     eng._p(scale), eng._p(shift), None, eng._p(y), DS_EPI_AFFINE | DS_EPI_CLIP | DS_EPI_OUT_F32, st)
    elif arith == "bf16x3":
        hi, lo = eng._pack_bf16(wt, 3)
        eng.lib.call("ds_conv_fwd_bf16", ctypes.byref(shp), eng._p(torch.from_numpy(x).cuda()), eng._p(hi), eng._p(lo),
                     eng._p(scale), eng._p(shift), None, eng._p(y), None, DS_EPI_AFFINE | DS_EPI_CLIP, st)
    else:
        wp = torch.empty(wt.numel(), device="cuda")
        eng.lib.call("ds_pack_conv_weight_f32", eng._p(wt), eng._p(wp), C, C, 3, 0, st)
        eng.lib.call("ds_conv_fwd_f32", ctypes.byref(shp), eng._p(torch.from_numpy(x).cuda()), eng._p(wp), eng._p(scale),
                     eng._p(shift), None, eng._p(y), None, DS_EPI_AFFINE | DS_EPI_CLIP, st)
    got = y[0, 1, :, 3].cpu().numpy()
    np.testing.assert_array_equal(got, np.clip(vals, 0.0, 20.0))
    assert float(y.min()) == 0.0 and float(y.max()) == 20.0


def test_clip_mask_is_strict_in_backward():
    """Hardtanh(0, 20) backward (model.py:36-44): gradient 1 iff 0 < x < 20 -- exactly 0 and exactly 20 are masked.
    Checked on the BatchNorm-backward reduce kernel, which applies the mask of the saved activation."""
    from deepspeaker_pytorch_amd.model import get_engine
    eng = get_engine()
    C, n_pix = 64, 256
    act = torch.full((n_pix, C), 5.0).cuda()
    special = [0.0, 20.0, float(np.nextafter(np.float32(0), np.float32(1))), float(np.nextafter(np.float32(20), np.float32(0)))]
    for k, v in enumerate(special):
        act[k, :] = v
    g1 = torch.ones((n_pix, C)).cuda()
    z = torch.randn((n_pix, C)).cuda()
    mean, invstd = torch.zeros(C).cuda(), torch.ones(C).cuda()
    gy = torch.empty_like(g1)
    rows = eng.lib.raw("ds_bn_bwd_partial_rows")(n_pix, C)
    partial = torch.empty((rows, C, 2)).cuda()
    eng.lib.call("ds_bn_bwd_reduce_f32", eng._p(g1), None, eng._p(act), eng._p(z), eng._p(mean), eng._p(invstd),
                 eng._p(gy), eng._p(partial), n_pix, C, eng._stream(gy))
    got = gy[:6, 0].cpu().numpy()
    np.testing.assert_array_equal(got, np.array([0.0, 0.0, 1.0, 1.0, 1.0, 1.0], np.float32))


@pytest.mark.parametrize("precision,bar", [("f32", 1e-3), ("bf16x3", 3e-3)])
def test_whole_network_gradients_on_the_clean_fixture(golden, precision, bar):
    """A fixture chosen (tests/golden/make_golden.py: find_clean_seed) so that no clipped-ReLU input of the reference
    lies within 3e-5 of 0 or 20: the fp32 / split-bf16 forward takes the reference's masks, and EVERY parameter
    gradient of the whole network is held to 1e-3 (f32) / 3e-3 (split bf16) of the float64 reference digest --
    not the 8e-2 band the mask-flipping fixture needs."""
    seed = int(golden["tight_seed_margin"][0])
    sd = O.make_state_dict(seed=seed, num_classes=16)
    x = torch.from_numpy(O.make_input(seed=seed + 1000, batch=2, frames=16)).cuda()
    m = build(sd, precision).train()
    e = m(x)
    # (train-mode BatchNorm over 2 x 1 x 4 pixels at the last stage amplifies rounding: looser than the eval bars)
    assert rel_err(e.detach().cpu().numpy(), golden["tight64_emb"]) < (1e-5 if precision == "f32" else 2e-4)
    ge = np.random.RandomState(78).randn(2, 512).astype(np.float32)
    m.zero_grad()
    e.backward(torch.from_numpy(ge).cuda())
    worst = {}
    for name, p in m.named_parameters():
        if name.startswith("model.classifier"):
            continue
        d64 = golden["tight64_grad/" + name]
        dg = grad_digest(p.grad.cpu().numpy())
        worst[name] = float(np.abs(dg - d64).max() / np.abs(d64).max())
    print("\n", precision, "worst gradient digest error:", max(worst.values()), max(worst, key=worst.get))
    assert max(worst.values()) < bar, worst


def test_classifier_head_gradients_vs_reference(golden):
    """model.forward_classifier + CrossEntropyLoss + backward (train_triplet.py:277-291) against the reference's own
    logits, loss and gradients: classifier weight / bias in full, fc and conv4 by digest."""
    from deepspeaker_pytorch_amd.model import CrossEntropyLoss
    sd = O.make_state_dict(seed=31, num_classes=16)
    m = build(sd).train()
    x = torch.from_numpy(O.make_input(seed=91, batch=6)).cuda()
    labels = torch.from_numpy(golden["cls_labels"]).cuda()
    logits = m.forward_classifier(x)
    assert rel_err(logits.detach().cpu().numpy(), golden["cls_logits"]) < 1e-4
    ce = CrossEntropyLoss().forward(logits, labels)
    assert abs(float(ce) - float(golden["cls_loss"])) < 1e-5 * max(1.0, abs(float(golden["cls_loss"])))
    m.zero_grad()
    ce.backward()
    assert rel_err(m.model.classifier.weight.grad.cpu().numpy(), golden["cls_grad_weight"]) < 1e-4
    assert rel_err(m.model.classifier.bias.grad.cpu().numpy(), golden["cls_grad_bias"]) < 1e-4
    for name, key in (("fc", "cls_grad_fc_digest"), ("conv4", "cls_grad_conv4_digest")):
        dg = grad_digest(getattr(m.model, name).weight.grad.cpu().numpy())
        ref = golden[key]
        assert np.abs(dg - ref).max() <= 5e-2 * np.abs(ref).max(), name     # (one clip mask may flip: see test_gpu_parity)


@pytest.mark.parametrize("precision", ["f32", "f16"])
def test_variable_length_batches_vs_reference(golden, precision):
    """BASELINE configs[4] through the path the tool uses (DeepSpeakerModel.embed_variable_length): utterances of 100,
    137 (not a multiple of anything), 237, 402 and 800 frames mixed into zero-padded batches must reproduce the
    reference's embedding of each utterance alone, and -- bitwise -- the model's own single-utterance forward."""
    from deepspeaker_pytorch_amd import scoring
    sd = O.make_state_dict(seed=11, num_classes=16)
    m = build(sd, precision).eval()
    utts, refs = [], []
    for T in (402, 100, 800, 137, 237):
        x = O.make_input(seed=100 + T, batch=2, frames=T)
        for b in range(2):
            utts.append(torch.from_numpy(x[b, 0]).cuda())
            refs.append(golden[f"full_eval_T{T}_emb"][b])
    with torch.no_grad():
        e = m.embed_variable_length(utts, max_batch=4)
        assert rel_err(e.cpu().numpy(), np.stack(refs)) < TOL[precision]
        for i in (1, 6, 4):
            alone = m(utts[i].reshape(1, 1, -1, 64))
            assert torch.equal(e[i:i + 1], alone)
        # the same utterances from a device-resident corpus (one gather kernel per batch)
        from deepspeaker_pytorch_amd.data import FeatureStore
        assert torch.equal(m.embed_variable_length(FeatureStore([u.cpu().numpy() for u in utts]), max_batch=4), e)
        # enrolment: the mean of the distances to a speaker's utterances (train_triplet.py:348-350)
        sc = scoring.enrolment_scores(e[:2], e[2:], [5, 3])
    en = e[2:].cpu().numpy()
    want = [O.pairwise_distance(np.repeat(e[0:1].cpu().numpy(), 5, 0), en[:5]).mean(),
            O.pairwise_distance(np.repeat(e[1:2].cpu().numpy(), 3, 0), en[5:]).mean()]
    assert np.abs(sc.cpu().numpy() - np.array(want)).max() < 1e-4


def test_low_latency_split_k_forward(golden):
    """DeepSpeakerModel(low_latency=True): small fp16 launches split their contraction over workgroups.  Same result
    as the one-pass path up to the f32 summation order, inside the contract against the reference, deterministic."""
    sd = O.make_state_dict(seed=11, num_classes=16)
    from deepspeaker_pytorch_amd.model import DeepSpeakerModel
    x = torch.from_numpy(O.make_input(seed=12, batch=6)).cuda()
    outs = {}
    for low in (False, True):
        m = DeepSpeakerModel(512, 16, precision="f16", low_latency=low)
        m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
        m = m.cuda().eval()
        with torch.no_grad():
            outs[low] = (m(x).clone(), m(x[:1].contiguous()).clone(), m(x).clone())
    assert torch.equal(outs[True][0], outs[True][2])
    # (a last-bit difference of an f32 sum can flip the fp16 rounding of an activation: fp16-noise-level agreement)
    assert rel_err(outs[True][0].cpu().numpy(), outs[False][0].cpu().numpy()) < 5e-4
    assert rel_err(outs[True][1].cpu().numpy(), outs[False][1].cpu().numpy()) < 5e-4
    assert rel_err(outs[True][0].cpu().numpy(), golden["full_eval_emb"]) < TOL["f16"]
