"""The fp16 throughput path OFF the distribution its tolerance and near-tie band were first measured on (N(0,1) inputs,
random-init weights, 160 frames) -- VERDICT r3 "next" #1.  On a real MI355X, at the configs[1] batch:

 (a) inputs x15 (real features are 20*log10 mel energies with std ~ 10-20),
 (c) 100- and 800-frame utterances,
     -- against outputs of the UNMODIFIED reference (tests/golden/reference_offdist.npz, made by make_golden.py);
 (b) weights after 24 fused-Adagrad training steps on speaker-structured synthetic features of realistic scale
     (BatchNorm statistics and clip occupancy have moved; embeddings are spread apart, so distances -- and the absolute
     error of a distance -- are larger) -- against the CPU oracle (oracle/torch_restatement.py, bit-pinned to the
     reference forward) evaluated on the very weights the GPU trained.

Asserted in each: embeddings and loss within north_star's 1e-3, the refined selection IDENTICAL to the reference's, and
the fp16 error of the filter's decision variable d_n - d_p below 0.75 x the band the call used.  Plus: a planted
too-narrow band is detected by the call's own probes, and a trained network makes the measured band grow."""
import os

import numpy as np
import pytest
import torch

import deepspeaker_oracle as O
from conftest import ROOT, rel_err

pytestmark = pytest.mark.gpu

CONTRACT = 1e-3
EMB_BAR = {"f32": 2e-5, "bf16x3": 4e-5, "f16": CONTRACT, "f16raw": CONTRACT}
CASES = {"x15": (4321, 768, 160, 15.0), "T100": (4322, 768, 100, 1.0), "T800": (4323, 384, 800, 1.0),
         "x15_T800": (4324, 96, 800, 15.0)}          # == tests/golden/make_golden.py OFFDIST_CASES


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(ROOT, "tests", "golden", "reference_offdist.npz"))


def build(sd, precision, num_classes=1211):
    """precision "f16raw": the fp16 path with its precision guard switched off (f16_guard=None)"""
    from deepspeaker_pytorch_amd.model import DeepSpeakerModel
    kw = {"f16_guard": None} if precision == "f16raw" else {}
    m = DeepSpeakerModel(512, num_classes, precision="f16" if precision == "f16raw" else precision, **kw)
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    return m.cuda().eval()


def case_inputs(name, gold):
    seed, rows, frames, scale = CASES[name]
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.randn(rows, 1, frames, 64, generator=g) * scale
    dig = gold[f"{name}_input_digest"]
    assert abs(float(x.double().sum()) - dig[0]) < 1e-6 * max(1.0, abs(dig[0])) and float(x[-1, 0, -1, -1]) == dig[2]
    return x.cuda()


def run_case(m, x, precision, ref_emb, ref_dp, ref_dn, ref_loss, ref_sel, tag, emb_bar=None):
    from deepspeaker_pytorch_amd.mining import select_triplets
    from deepspeaker_pytorch_amd.model import TripletMarginLoss
    nt = x.shape[0] // 3
    with torch.no_grad():
        e = m(x).clone()
        a, p, n = e[:nt], e[nt:2 * nt], e[2 * nt:]
        loss = TripletMarginLoss(0.1).forward(a, p, n)
        raw = select_triplets(a, p, n, 0.1)                                   # the path's own distances, unrefined
        sel = select_triplets(a, p, n, 0.1, model=m, inputs=(x[:nt], x[nt:2 * nt], x[2 * nt:]))
    err = rel_err(e.cpu().numpy(), ref_emb)
    # ... and the worst ROW's relative L2 error, as tests/test_gpu_bench_size.py asserts (max |d| / max |ref| alone lets a
    # row whose every element is off by a little pass)
    ref64 = torch.from_numpy(np.asarray(ref_emb)).double()
    row = float(((e.cpu().double() - ref64).norm(dim=1) / ref64.norm(dim=1)).max())
    gap_ref = ref_dn - ref_dp - np.float32(0.1)
    gap_raw = raw.d_n.cpu().numpy() - raw.d_p.cpu().numpy() - np.float32(0.1)
    gap_err = float(np.abs(gap_raw - gap_ref).max())
    loss_rel = abs(float(loss) - float(ref_loss)) / max(abs(float(ref_loss)), 1e-6)
    obs = sel.observed_error
    print(f"\n[{tag} {precision}] emb max|d|/max {err:.3e}, worst row rel-L2 {row:.3e}; loss rel {loss_rel:.3e}; max |error of d_n - d_p| {gap_err:.3e}; "
          f"band used {sel.band:.3e}; near ties {sel.n_near_ties} in {sel.amb_cap} slots; probes saw {obs[0]} over {obs[1]} "
          f"slots; fell back: {sel.refined_all}; reference min |gap| {np.abs(gap_ref).min():.3e}, mean d_n "
          f"{float(ref_dn.mean()):.3f}")
    assert err < (emb_bar or EMB_BAR[precision]), err
    assert row < (emb_bar or EMB_BAR[precision]), row
    assert loss_rel < CONTRACT
    guard = getattr(m, "f16_guard", None)
    on_f16 = precision in ("f16", "f16raw") and (guard is None or guard.verdict == "f16")      # fp16 kernels produced `e`
    if guard is not None:
        print(f"    precision guard: {guard.report()}")
    if on_f16 and sel.embedding_error is not None:
        # the path's own estimate of its embedding error (fp16 vs f32-class on the 3 x slots sampled rows) is of the size
        # of the true error over all rows: the same measure on a sample can only be smaller, and not by much
        print(f"    embedding error the call observed on its {3 * sel.amb_cap} sampled rows: {sel.embedding_error:.3e} (all rows: {err:.3e})")
        assert 0.3 * err < sel.embedding_error < 1.2 * err, (sel.embedding_error, err)
    np.testing.assert_array_equal(sel.indices.cpu().numpy(), ref_sel)         # identical selection
    if on_f16:
        # the band covers what fp16 does -- or the call's own probes noticed that it does not and it fell back
        assert gap_err < 0.75 * sel.band or (sel.band_exceeded and sel.refined_all), (gap_err, sel.band)
    return gap_err, sel


@pytest.mark.parametrize("precision", ["f32", "bf16x3", "f16"])
@pytest.mark.parametrize("name", list(CASES))
def test_offdist_inputs_vs_reference_golden(gold, name, precision):
    sd = O.make_state_dict(seed=0, num_classes=1211)
    m = build(sd, precision)
    x = case_inputs(name, gold)
    run_case(m, x, precision, gold[f"{name}_emb"], gold[f"{name}_d_p"], gold[f"{name}_d_n"], gold[f"{name}_loss"],
             gold[f"{name}_selected"], name)


def _train_on_speakers(optimizer="adagrad", steps=24, lr=0.01, triplets=32, speakers=16, utts=8):
    """`steps` fused-Adagrad steps (train_triplet.py:369-383, lr_decay 1e-4; lr 0.01: at the script's default of 0.1
    Adagrad's sign-like first step moves every filter by 0.1 -- three times the init's standard deviation -- and 24
    steps later the network saturates at 0 / 20 almost everywhere, values fp16 holds exactly: no test) of the triplet regime
    (train_triplet.py:215-224) on speaker-structured features of realistic scale, f32 arithmetic; returns the trained
    state_dict (CPU tensors) and the corpus."""
    from deepspeaker_pytorch_amd.model import DeepSpeakerModel, TripletMarginLoss
    from deepspeaker_pytorch_amd.optim import create_optimizer
    sd = O.make_state_dict(seed=5, num_classes=speakers, randomize_bn=False)
    m = DeepSpeakerModel(512, speakers, precision="f32")
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    m = m.cuda().train()
    opt = create_optimizer(m, lr, optimizer, lr_decay=1e-4)       # "sgd": momentum 0.9, dampening 0.9 (train_triplet.py:372-374)
    corpus = O.make_speaker_corpus(31, speakers, utts, 160, scale=12.0, mix=(0.4, 0.3, 0.85))
    losses = []
    for it in range(steps):
        a, p, n, _, _ = O.sample_triplets(500 + it, speakers, utts, triplets)
        xs = [torch.from_numpy(O.gather_utterances(corpus, i)).cuda() for i in (a, p, n)]
        oa, op, on = m.forward_triplet(*xs)
        loss = TripletMarginLoss(0.1).forward(oa, op, on)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    torch.cuda.synchronize()
    return {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}, corpus, losses


@pytest.fixture(scope="module", params=[("adagrad", 0.01, 24), ("sgd", 0.02, 40)], ids=["adagrad", "sgd"])
def trained(request):
    import torch_restatement as TR
    opt_name, lr, steps = request.param
    tsd, corpus, losses = _train_on_speakers(opt_name, steps=steps, lr=lr)
    # the evaluation batch: 256 triplets of the same corpus (the reference's validation works on training speakers too)
    a, p, n, _, _ = O.sample_triplets(9000, corpus.shape[0], corpus.shape[1], 256)
    x = torch.from_numpy(np.concatenate([O.gather_utterances(corpus, i) for i in (a, p, n)]))
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    with torch.no_grad():
        ref = torch.cat([TR.forward_eval(tsd, x[i:i + 64]) for i in range(0, 768, 64)]).numpy()
    ra, rp, rn = ref[:256], ref[256:512], ref[512:]
    ref_loss, d_p, d_n = O.triplet_margin_loss(ra, rp, rn, 0.1)
    ref_sel, _, _ = O.triplet_filter(d_p, d_n, 0.1)
    return tsd, x, ref, d_p.astype(np.float32), d_n.astype(np.float32), ref_loss, ref_sel, losses


@pytest.mark.parametrize("precision", ["f32", "bf16x3", "f16", "f16raw"])
def test_trained_weights_realistic_inputs_vs_oracle(trained, precision, request):
    """(b): the weights are whatever 24 Adagrad / 40 SGD steps on the GPU made of them; the checker is the CPU oracle on
    those same weights.  BatchNorm running statistics now match inputs of std 12, the clip is hit from above, distances are
    no longer the sub-unit ones of a random-init network.

    The SGD-trained network spreads the embeddings (mean d_n 6.0 against 1.3 at random init) and RAW fp16 then sits AT the
    1e-3 contract (measured 0.86e-3 - 1.005e-3; every layer contributes, tools/f16_error_budget.py) with an error of
    d_n - d_p (2.2e-3) past the band's floor.  precision="f16" (the default, guarded) must be INSIDE the contract on both
    networks with no widened bar: its guard measures the first rows of the first forward against the f32-class path and
    runs the f32-class kernels on the SGD network.  "f16raw" (f16_guard=None) is the unguarded arithmetic, kept to 1.5e-3
    and to the band machinery that protects its selection."""
    tsd, x, ref, d_p, d_n, ref_loss, ref_sel, losses = trained
    print("\ntraining losses:", " ".join(f"{v:.3f}" for v in losses))
    m = build({k: v.numpy() for k, v in tsd.items()}, precision, num_classes=tsd["model.classifier.bias"].numel())
    bar = {"f16": CONTRACT, "f16raw": 1.5e-3}.get(precision, 5e-5)
    gap_err, sel = run_case(m, x.cuda(), precision, ref, d_p, d_n, ref_loss, ref_sel, "trained", emb_bar=bar)
    if precision == "f16":
        g = m.f16_guard
        sgd = "sgd" in request.node.callspec.id
        assert g.checks == 1 and g.sample_error is not None
        # what the guard measured on 32 rows is of the size of the raw path's true error over all rows (printed by the
        # "f16raw" case); its estimate put the SGD network above the threshold and the Adagrad one far below
        assert (g.verdict == "bf16x3" and g.estimate > g.threshold) if sgd else (g.verdict == "f16" and g.estimate < 0.2 * g.threshold)
    if precision == "f16raw":
        from deepspeaker_pytorch_amd.mining import REFINE_BAND, refine_policy, select_triplets
        # several more calls: the policy's band settles at what this network's fp16 error needs, and stays sufficient
        xd = x.cuda()
        with torch.no_grad():
            e = m(xd).clone()
            for _ in range(6):
                s2 = select_triplets(e[:256], e[256:512], e[512:], 0.1, model=m, inputs=(xd[:256], xd[256:512], xd[512:]))
                np.testing.assert_array_equal(s2.indices.cpu().numpy(), ref_sel)
        pol = refine_policy(m)
        print(f"policy after 7 calls: band {pol.band_for():.3e} (floor {REFINE_BAND:.3e}), observed max {pol.err_max_window:.3e} "
              f"over {pol.err_samples} slots, violations {pol.band_violations}, overflows {pol.overflows}")
        assert pol.band_for() >= 2.0 * min(gap_err, pol.err_max_window) or pol.band_for() == REFINE_BAND
        assert gap_err < 0.75 * pol.band_for()              # after the first calls the band is what this network needs
        assert s2.band == pol.band_for() or s2.band >= 2.0 * pol.err_max_window
        assert not s2.band_exceeded


def test_planted_too_narrow_band_is_detected(gold):
    """A band far below the fp16 error (1e-5 against ~5e-4): the call's own slots -- near ties or probes -- see an error
    above half of it, the selection falls back to the whole batch at f32-class precision and is still the reference's."""
    from deepspeaker_pytorch_amd.mining import refine_policy, select_triplets
    g1 = np.load(os.path.join(ROOT, "tests", "golden", "reference_cfg1.npz"))
    x = torch.randn(768, 1, 160, 64, generator=torch.Generator(device="cpu").manual_seed(1234)).cuda()
    m = build(O.make_state_dict(seed=0, num_classes=1211), "f16")
    with torch.no_grad():
        e = m(x).clone()
        sel = select_triplets(e[:256], e[256:512], e[512:], 0.1, model=m, inputs=(x[:256], x[256:512], x[512:]), band=1e-5)
        idx = sel.indices.cpu().numpy()
    assert sel.band_exceeded and sel.refined_all and not sel.refine_overflow
    assert refine_policy(m).band_violations == 1
    np.testing.assert_array_equal(idx, g1["cfg1_selected"])
    # and the next call, left to the policy, uses a band that covers what was seen
    with torch.no_grad():
        nxt = select_triplets(e[:256], e[256:512], e[512:], 0.1, model=m, inputs=(x[:256], x[256:512], x[512:]))
    assert nxt.band >= 2.0 * sel.observed_error[0] and not nxt.band_exceeded
    np.testing.assert_array_equal(nxt.indices.cpu().numpy(), g1["cfg1_selected"])


def test_guard_rechecks_without_a_host_synchronisation_and_follows_the_weights(gold):
    """precision_guard.F16Guard on the GPU: (a) the periodic re-check runs on the side stream and is taken in by a later
    forward (no synchronisation: `rechecks` counts, `checks` does not), embeddings unchanged bit for bit; (b) weights that
    make fp16 inadequate (every filter scaled so that activations sit at the top of fp16's coarse range is not needed: a
    threshold below the measured error does it) escalate, and the forward then returns the f32-class path's embeddings;
    (c) a capture sees the standing verdict and measures nothing."""
    from deepspeaker_pytorch_amd.model import DeepSpeakerModel
    sd = O.make_state_dict(seed=0, num_classes=16)
    x = case_inputs("x15", gold)[:96]
    m = build(sd, "f16", num_classes=16)
    g = m.f16_guard
    g.recheck = 3
    with torch.no_grad():
        e0 = m(x).clone()
        assert g.checks == 1 and g.verdict == "f16" and g.rechecks == 0
        for _ in range(8):
            assert torch.equal(m(x), e0)
        torch.cuda.synchronize()
        m(x)
    assert g.checks == 1 and g.rechecks >= 2 and g.source in ("check", "recheck") and g.verdict == "f16"
    assert abs(g.sample_error - 4.2e-4) < 1.5e-4            # the re-check read the same thing the first check did
    # (b) a guard that cannot be met: the same weights, threshold below the measured error
    m2 = DeepSpeakerModel(512, 16, precision="f16", f16_guard=1e-4)
    m2.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    m2 = m2.cuda().eval()
    m3 = build(sd, "bf16x3", num_classes=16)
    with torch.no_grad():
        e2, e3 = m2(x).clone(), m3(x).clone()
    assert m2.f16_guard.verdict == "bf16x3" and m2.f16_guard.escalations == 1 and torch.equal(e2, e3)
    # (c) inside a capture nothing is measured
    m4 = build(sd, "f16", num_classes=16)
    gr = m4.graphed(x)                                    # warm-up forwards (outside the capture) run the check
    assert m4.f16_guard.checks == 1
    n_calls = m4.f16_guard.calls
    assert torch.equal(gr(x), e0) and m4.f16_guard.calls == n_calls and m4.f16_guard.checks == 1
