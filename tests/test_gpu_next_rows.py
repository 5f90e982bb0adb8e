"""GPU checks of the SURVEY 8(f) "next" rows and the softmax head (a12): fused optimizers, device-side
batch assembly, verification scoring + ROC/EER, classifier GEMM + cross-entropy."""
import numpy as np
import pytest
import torch

import deepspeaker_oracle as O
from conftest import perf_note, rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind", ["adagrad", "sgd", "sgd_plain", "adam"])
def test_fused_optimizer_vs_torch(kind):
    from deepspeaker_pytorch_amd import optim as fo
    rs = np.random.RandomState(3)
    shapes = [(64, 1, 5, 5), (17,), (33000,), (512, 512, 3, 3)]
    ref_p = [torch.nn.Parameter(torch.from_numpy(rs.randn(*s).astype(np.float32)).cuda()) for s in shapes]
    our_p = [torch.nn.Parameter(p.detach().clone()) for p in ref_p]
    if kind == "adagrad":       # train_triplet.py:379-382
        ref = torch.optim.Adagrad(ref_p, lr=0.1, lr_decay=1e-4, weight_decay=1e-3)
        ours = fo.FusedAdagrad(our_p, lr=0.1, lr_decay=1e-4, weight_decay=1e-3)
    elif kind == "sgd":         # train_triplet.py:372-374
        ref = torch.optim.SGD(ref_p, lr=0.1, momentum=0.9, dampening=0.9, weight_decay=1e-3)
        ours = fo.FusedSGD(our_p, lr=0.1, momentum=0.9, dampening=0.9, weight_decay=1e-3)
    elif kind == "sgd_plain":   # no momentum: the optimizer has no state tensors at all (a null state table)
        ref = torch.optim.SGD(ref_p, lr=0.05)
        ours = fo.FusedSGD(our_p, lr=0.05)
    else:                       # train_triplet.py:376-377
        ref = torch.optim.Adam(ref_p, lr=0.01, weight_decay=1e-3)
        ours = fo.FusedAdam(our_p, lr=0.01, weight_decay=1e-3)
    for it in range(4):
        for a, b in zip(ref_p, our_p):
            g = torch.from_numpy(rs.randn(*a.shape).astype(np.float32)).cuda()
            a.grad, b.grad = g.clone(), g.clone()
        ref.step()
        ours.step()
    for a, b in zip(ref_p, our_p):
        assert rel_err(b.detach().cpu().numpy(), a.detach().cpu().numpy()) < 3e-6
    ref.load_state_dict(ours.state_dict())


def test_training_with_fused_adagrad_matches_torch_adagrad():
    from deepspeaker_pytorch_amd.model import DeepSpeakerModel, TripletMarginLoss
    from deepspeaker_pytorch_amd.optim import create_optimizer
    sd = O.make_state_dict(seed=31, num_classes=16)
    xs = [torch.from_numpy(O.make_input(seed=32 + i, batch=4)).cuda() for i in range(3)]
    finals = []
    for fused in (False, True):
        m = DeepSpeakerModel(512, 16)
        m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
        m = m.cuda().train()
        opt = create_optimizer(m, 0.1) if fused else torch.optim.Adagrad(m.parameters(), lr=0.1, lr_decay=1e-4)
        # ONE step: Adagrad's first update is lr * g / (|g| + 1e-10), a sign step, so a second step would
        # amplify last-bit differences between the two optimizers' arithmetic into O(lr) differences on every
        # weight whose gradient is at rounding-noise level -- a property of Adagrad, not of either code
        loss = TripletMarginLoss(0.1).forward(m(xs[0]), m(xs[1]), m(xs[2]))
        opt.zero_grad()
        loss.backward()
        opt.step()
        finals.append({k: v.detach().clone() for k, v in m.state_dict().items()})
    for k in finals[0]:
        if finals[0][k].dtype.is_floating_point:
            assert rel_err(finals[1][k].cpu().numpy(), finals[0][k].cpu().numpy()) < 1e-5, k


def test_feature_store_and_scoring(golden):
    from deepspeaker_pytorch_amd import data, scoring
    rs = np.random.RandomState(4)
    utts = [rs.randn(t, 64).astype(np.float32) for t in (500, 333, 1200)]
    fs = data.FeatureStore(utts)
    x = fs.crops([2, 0, 1], [100, 340, 0], 160)
    assert x.shape == (3, 1, 160, 64) and x.is_cuda
    np.testing.assert_array_equal(x[0, 0].cpu().numpy(), utts[2][100:260])
    np.testing.assert_array_equal(x[1, 0, :160].cpu().numpy()[:160], utts[0][340:500])
    v = scoring.evaluate(torch.from_numpy(golden["roc_dist"]).cuda(), torch.from_numpy(golden["roc_labels"]).cuda())
    np.testing.assert_allclose([v.tpr, v.fpr, v.accuracy], golden["roc_tpr_fpr_acc"], atol=1e-6)
    tp, fp, best, *_ = O.roc_sweep(golden["roc_dist"], golden["roc_labels"], np.arange(0, 30, 0.01))
    np.testing.assert_array_equal(v.tp.cpu().numpy(), tp)
    issame = golden["roc_labels"].astype(bool)
    assert abs(v.eer - O.equal_error_rate(tp, fp, issame.sum(), (~issame).sum())) < 1e-5
    a, p = rs.randn(64, 512).astype(np.float32), rs.randn(64, 512).astype(np.float32)
    s = scoring.trial_scores(torch.from_numpy(a).cuda(), torch.from_numpy(p).cuda(), 8)
    assert rel_err(s.cpu().numpy(), O.test_scores(a, p, 8)) < 1e-6


def test_softmax_pretraining_regime(golden):
    """train_triplet.py:277-291: logits of cat[a,p,n] through forward_classifier, CE + loss_ratio * triplet,
    backward reaches the classifier AND the network."""
    from deepspeaker_pytorch_amd.model import CrossEntropyLoss, DeepSpeakerModel, TripletMarginLoss
    sd = O.make_state_dict(seed=31, num_classes=16)
    m = DeepSpeakerModel(512, 16)
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    m = m.cuda().train()
    xs = [torch.from_numpy(O.make_input(seed=32 + i, batch=4)).cuda() for i in range(3)]
    cls = [m.forward_classifier(x) for x in xs]
    labels = torch.tensor([1, 2, 3, 4, 1, 2, 3, 4, 5, 6, 7, 8]).cuda()
    logits = torch.cat(cls)
    ce = CrossEntropyLoss().forward(logits, labels)
    ref_ce = torch.nn.functional.cross_entropy(logits.detach(), labels)
    assert abs(float(ce) - float(ref_ce)) < 1e-5
    ce.backward()
    assert m.model.classifier.weight.grad is not None and m.model.conv1.weight.grad is not None
    g = m.model.classifier.weight.grad
    assert torch.isfinite(g).all() and float(g.abs().max()) > 0
    v = CrossEntropyLoss().forward(torch.from_numpy(golden["ce_logits"]).cuda(), torch.from_numpy(golden["ce_labels"]).cuda())
    assert abs(float(v) - float(golden["ce_value"])) < 1e-6


def test_graph_replay_matches_eager():
    """HIP-graph replay of the eval forward (DeepSpeakerModel.graphed) is bit-identical to eager launches."""
    from deepspeaker_pytorch_amd.model import DeepSpeakerModel
    sd = O.make_state_dict(seed=3, num_classes=8)
    m = DeepSpeakerModel(512, 8, precision="bf16x3")
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    m = m.cuda().eval()
    x = torch.from_numpy(O.make_input(seed=11, batch=4)).cuda()
    g = m.graphed(x)
    with torch.no_grad():
        ref = m(x).clone()
        assert torch.equal(g(x), ref)
        x2 = torch.from_numpy(O.make_input(seed=12, batch=4)).cuda()
        assert torch.equal(g(x2), m(x2))


def test_softmax_regime_reuses_embeddings():
    """the pre-training regime of train_triplet.py:277-287 without the three extra network forwards.
    In eval-BatchNorm terms `classify(out[idx])` must equal `forward_classifier(data[idx])` (rows are independent),
    the fused `classifier_loss` must equal CrossEntropyLoss over those logits, gradients reach the classifier and the
    embeddings, and the step is shorter by about three forwards."""
    import time
    from deepspeaker_pytorch_amd.mining import select_triplets
    from deepspeaker_pytorch_amd.model import CrossEntropyLoss, DeepSpeakerModel
    sd = O.make_state_dict(seed=31, num_classes=16)
    m = DeepSpeakerModel(512, 16, precision="bf16x3")
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    m = m.cuda().eval()
    B = 64
    data = [torch.from_numpy(O.make_input(seed=60 + i, batch=B)).cuda() for i in range(3)]
    rs = np.random.RandomState(9)
    label_p, label_n = (torch.from_numpy(rs.randint(0, 16, B).astype(np.int64)).cuda() for _ in range(2))

    def reference_style():
        with torch.no_grad():
            outs = [m(x) for x in data]                                        # train_triplet.py:215
            idx = select_triplets(*outs, margin=0.1).indices                   # :251-262
        cls = [m.forward_classifier(x[idx]) for x in data]                     # :277-279 (three more forwards)
        labels = torch.cat([label_p[idx], label_p[idx], label_n[idx]])         # :283-284
        return CrossEntropyLoss().forward(torch.cat(cls), labels), idx

    def reuse_style():
        with torch.no_grad():
            outs = [m(x) for x in data]
            idx = select_triplets(*outs, margin=0.1).indices
        emb = torch.cat([o[idx] for o in outs]).requires_grad_(True)
        labels = torch.cat([label_p[idx], label_p[idx], label_n[idx]])
        return m.classifier_loss(emb, labels), emb, labels

    l_ref, idx = reference_style()
    l_new, emb, labels = reuse_style()
    assert idx.numel() > 0
    assert abs(float(l_ref) - float(l_new)) < 1e-5 * max(1.0, abs(float(l_ref)))
    logits = m.classify(emb.detach())
    assert abs(float(CrossEntropyLoss().forward(logits, labels)) - float(l_new)) < 1e-6
    m.zero_grad()
    l_new.backward()
    gw = m.model.classifier.weight.grad.clone()
    ref_w = m.model.classifier.weight.detach().clone().requires_grad_(True)
    ref_b = m.model.classifier.bias.detach().clone().requires_grad_(True)
    ref_e = emb.detach().clone().requires_grad_(True)
    torch.nn.functional.cross_entropy(torch.nn.functional.linear(ref_e, ref_w, ref_b), labels).backward()
    assert rel_err(gw.cpu().numpy(), ref_w.grad.cpu().numpy()) < 1e-5
    assert rel_err(emb.grad.cpu().numpy(), ref_e.grad.cpu().numpy()) < 1e-5
    # the packed classifier copies are built once per parameter version
    assert m._head() is m._head()

    def timeit(fn, n=10):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    t_ref, t_new = timeit(reference_style), timeit(reuse_style)
    print(f"\nsoftmax regime step (B = {B}, eval BatchNorm): three extra forwards {t_ref:.2f} ms, re-used embeddings {t_new:.2f} ms")
    perf_note(t_new < t_ref, ("softmax regime: re-used embeddings vs three extra forwards", t_new, t_ref))


def test_overlapped_backward_same_gradients_and_memory():
    """Filter gradients on the second stream (backward._FilterGradLane): the same gradients bit for bit, and the caching
    allocator must not hold back the cross-stream blocks (Tensor.record_stream did: 3.7x the reserved memory)."""
    from deepspeaker_pytorch_amd import backward
    from deepspeaker_pytorch_amd.model import DeepSpeakerModel, TripletMarginLoss
    sd = O.make_state_dict(seed=41, num_classes=16)
    g = torch.Generator().manual_seed(5)
    xs = [torch.randn(48, 1, 160, 64, generator=g).cuda() for _ in range(3)]
    results = {}
    try:
        for overlap in (False, True):
            backward.OVERLAP_FILTER_GRADIENTS = overlap
            m = DeepSpeakerModel(512, 16, precision="bf16x3")
            m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
            m = m.cuda().train()
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
            for _ in range(4):
                m.zero_grad()
                TripletMarginLoss(0.1).forward(*m.forward_triplet(*xs)).backward()
            torch.cuda.synchronize()
            results[overlap] = ({k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None},
                                torch.cuda.memory_reserved())
            del m
    finally:
        backward.OVERLAP_FILTER_GRADIENTS = True
    (g0, r0), (g1, r1) = results[False], results[True]
    assert g0.keys() == g1.keys() and len(g0) > 30
    for k in g0:
        assert torch.equal(g0[k], g1[k]), k
    assert r1 < 1.5 * r0, (r0 >> 20, r1 >> 20)
