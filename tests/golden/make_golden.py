#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the UNMODIFIED reference
(/root/reference/model.py) on seeded inputs.  Runs only in the build container
(the reference is not present on the GPU box); the .npz outputs are committed.

    python tests/golden/make_golden.py

Inputs and parameters are NOT stored: they are regenerated from seeds by
oracle/deepspeaker_oracle.py (make_state_dict / make_input, frozen legacy
numpy RandomState streams), so each fixture holds only the reference's outputs.
"""
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, "/root/reference")
warnings.filterwarnings("ignore")

import deepspeaker_oracle as O          # noqa: E402
import model as ref                      # noqa: E402  (the reference, unmodified)

torch.set_num_threads(8)


def build_ref(sd_np, num_classes):
    m = ref.DeepSpeakerModel(512, num_classes)
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in sd_np.items()}
    m.load_state_dict(sd, strict=False)     # strict=False: small model lacks stages 3-4
    return m


def run_prefix(m, x, n_stages):
    """Stages 1..n_stages + pool + fc + l2norm*10, through the reference's own
    modules (model.py:185-213); n_stages=4 is exactly DeepSpeakerModel.forward."""
    r = m.model
    for i in range(1, n_stages + 1):
        x = getattr(r, f"conv{i}")(x)
        x = getattr(r, f"bn{i}")(x)
        x = r.relu(x)
        x = getattr(r, f"layer{i}")(x)
    x = r.avgpool(x)
    x = x.view(x.size(0), -1)
    x = r.fc(x)
    return m.l2_norm(x) * 10


TIGHT_SEED = 109            # first seed find_clean_seed() accepts (min distance to a clip boundary 3.02e-5)
TIGHT_MARGIN = 3e-5


def clip_margin(sd, x):
    """smallest distance of any clipped-ReLU input of the reference (float64, train mode) to a clip boundary"""
    m = build_ref(sd, 16).train().double()
    worst = [np.inf]

    def hook(_mod, inp):
        v = inp[0].detach()
        worst[0] = min(worst[0], float(torch.minimum(v.abs(), (v - 20).abs()).min()))

    hs = [mod.register_forward_pre_hook(hook) for mod in m.modules() if isinstance(mod, ref.ReLU)]
    m(torch.from_numpy(x).double())
    for h in hs:
        h.remove()
    return worst[0]


def clean_fixture():
    sd = O.make_state_dict(seed=TIGHT_SEED, num_classes=16)
    x = O.make_input(seed=TIGHT_SEED + 1000, batch=2, frames=16)
    margin = clip_margin(sd, x)
    assert margin > TIGHT_MARGIN, (TIGHT_SEED, margin)
    return sd, x, margin


def find_clean_seed(start=0, stop=2000):
    for seed in range(start, stop):
        sd = O.make_state_dict(seed=seed, num_classes=16)
        x = O.make_input(seed=seed + 1000, batch=2, frames=16)
        mg = clip_margin(sd, x)
        if mg > TIGHT_MARGIN:
            return seed, mg
    raise RuntimeError("no clean seed in range")


def grad_digest(t):
    a = t.detach().double().numpy().ravel()
    stride = max(1, a.size // 64)
    return np.concatenate([[np.sqrt((a * a).sum()), a.sum()], a[:16], a[::stride][:64]])


def main():
    out = {}
    # ---------------- full model, eval, T=160 (config 1/2 shape, small batch) ----
    sd = O.make_state_dict(seed=11, num_classes=16)
    m = build_ref(sd, 16).eval()
    x = O.make_input(seed=12, batch=6)
    with torch.no_grad():
        e = m(torch.from_numpy(x))
        cls = m.forward_classifier(torch.from_numpy(x))
    out["full_eval_emb"] = e.numpy()
    out["full_eval_cls"] = cls.numpy()
    # intermediate taps through hooks, for per-layer kernel checks
    with torch.no_grad():
        r = m.model
        t = torch.from_numpy(x)
        t = r.relu(r.bn1(r.conv1(t)))
        out["full_eval_stage1_a"] = t.numpy()[:1, :, :16].copy()
        t = r.layer1(t)
        out["full_eval_stage1_c"] = t.numpy()[:1, :, :16].copy()

    # ---------------- variable length, eval (config 4 shapes incl. odd sizes) ----
    for T in (100, 237, 402):
        xv = O.make_input(seed=100 + T, batch=2, frames=T)
        with torch.no_grad():
            out[f"full_eval_T{T}_emb"] = m(torch.from_numpy(xv)).numpy()

    # the ends of configs[4]'s length range and the shortest input the reference accepts
    for T in (1, 137, 800):
        xv = O.make_input(seed=100 + T, batch=2, frames=T)
        with torch.no_grad():
            out[f"full_eval_T{T}_emb"] = m(torch.from_numpy(xv)).numpy()
    # odd batch sizes (tile raggedness): one utterance, three, and one more than a power of two
    xb = O.make_input(seed=300, batch=257, frames=32)
    with torch.no_grad():
        out["full_eval_B257_T32_emb"] = m(torch.from_numpy(xb)).numpy()

    # ---------------- ResCNN-small (configs[0]): 2 stages, B=32 ------------------
    sds = O.make_state_dict(seed=21, num_classes=16, n_stages=2)
    ms = build_ref(sds, 16).eval()
    xs = O.make_input(seed=22, batch=32)
    with torch.no_grad():
        out["small_eval_emb"] = run_prefix(ms, torch.from_numpy(xs), 2).numpy()

    # ---------------- full model, train mode: fwd, stats, triplet loss, backward --
    sdt = O.make_state_dict(seed=31, num_classes=16)
    mt = build_ref(sdt, 16).train()
    B = 8
    xa, xp, xn = (O.make_input(seed=32 + i, batch=B) for i in range(3))
    ea, ep, en = mt(torch.from_numpy(xa)), mt(torch.from_numpy(xp)), mt(torch.from_numpy(xn))
    loss = ref.TripletMarginLoss(0.1).forward(ea, ep, en)
    mt.zero_grad()
    loss.backward()
    out["full_train_emb_a"] = ea.detach().numpy()
    out["full_train_emb_p"] = ep.detach().numpy()
    out["full_train_emb_n"] = en.detach().numpy()
    out["full_train_loss"] = loss.detach().numpy()
    for k, v in mt.state_dict().items():
        if "running" in k or "num_batches" in k:
            out["full_train_stat/" + k] = v.numpy()
    for k, p in mt.named_parameters():
        if p.grad is not None:
            out["full_train_grad/" + k] = grad_digest(p.grad)

    # single train-mode forward + backward from a fixed embedding gradient
    # (isolates the network backward from the loss backward)
    mt2 = build_ref(sdt, 16).train()
    e1 = mt2(torch.from_numpy(xa))
    ge = np.random.RandomState(77).randn(B, 512).astype(np.float32)
    mt2.zero_grad()
    e1.backward(torch.from_numpy(ge))
    out["single_train_emb"] = e1.detach().numpy()
    for k, p in mt2.named_parameters():
        if p.grad is not None:
            out["single_train_grad/" + k] = grad_digest(p.grad)
    for k, v in mt2.state_dict().items():
        if "running" in k or "num_batches" in k:
            out["single_train_stat/" + k] = v.numpy()

    # the same, with the reference evaluated in float64 (model.double()): the fp32 autograd
    # result above is itself ~1e-2 (elementwise, relative to max) away from this in the
    # random-init regime, so the fp64 run is the tight pin for backward restatements.
    mt3 = build_ref(sdt, 16).train().double()
    e3 = mt3(torch.from_numpy(xa).double())
    mt3.zero_grad()
    e3.backward(torch.from_numpy(ge).double())
    out["single_train64_emb"] = e3.detach().numpy()
    for k, p in mt3.named_parameters():
        if p.grad is not None:
            out["single_train64_grad/" + k] = grad_digest(p.grad)

    # ---------------- a fixture WITHOUT near-boundary clip inputs: tight whole-network gradients -------------
    # Every clipped ReLU of the reference receives values at least TIGHT_MARGIN away from 0 and 20 (float64 run),
    # so an fp32 / split-bf16 forward takes the same masks and whole-network gradients can be held to 1e-3.
    sd_c, x_c, margin_c = clean_fixture()
    out["tight_seed_margin"] = np.array([TIGHT_SEED, margin_c])
    for tag, dt in (("32", torch.float32), ("64", torch.float64)):
        mc = build_ref(sd_c, 16).train().to(dt)
        ec = mc(torch.from_numpy(x_c).to(dt))
        gec = np.random.RandomState(78).randn(*ec.shape).astype(np.float32)
        mc.zero_grad()
        ec.backward(torch.from_numpy(gec).to(dt))
        out[f"tight{tag}_emb"] = ec.detach().numpy()
        for k, p_ in mc.named_parameters():
            if p_.grad is not None:
                out[f"tight{tag}_grad/" + k] = grad_digest(p_.grad)

    # ---------------- softmax pre-training head with gradients (train_triplet.py:277-291) ----------------------
    mh = build_ref(sdt, 16).train()
    xh = torch.from_numpy(O.make_input(seed=91, batch=6))
    lab = torch.from_numpy(np.random.RandomState(92).randint(0, 16, 6).astype(np.int64))
    logits = mh.forward_classifier(xh)
    ce = torch.nn.CrossEntropyLoss()(logits, lab)
    mh.zero_grad()
    ce.backward()
    out["cls_logits"], out["cls_labels"], out["cls_loss"] = logits.detach().numpy(), lab.numpy(), ce.detach().numpy()
    out["cls_grad_weight"] = mh.model.classifier.weight.grad.numpy()
    out["cls_grad_bias"] = mh.model.classifier.bias.grad.numpy()
    out["cls_grad_fc_digest"] = grad_digest(mh.model.fc.weight.grad)
    out["cls_grad_conv4_digest"] = grad_digest(mh.model.conv4.weight.grad)

    # ---------------- loss side on free-standing embeddings ----------------------
    rs = np.random.RandomState(41)
    N = 96
    base = rs.randn(N, 512).astype(np.float32)
    a = (base / np.linalg.norm(base, axis=1, keepdims=True) * 10).astype(np.float32)
    p = a + rs.randn(N, 512).astype(np.float32) * 0.05
    n = a + rs.randn(N, 512).astype(np.float32) * 0.05
    ta, tp, tn = (torch.from_numpy(v).requires_grad_(True) for v in (a, p, n))
    pd = ref.PairwiseDistance(2)
    d_p, d_n = pd.forward(ta, tp), pd.forward(ta, tn)
    tl = ref.TripletMarginLoss(0.1).forward(ta, tp, tn)
    tl.backward()
    out["loss_d_p"], out["loss_d_n"] = d_p.detach().numpy(), d_n.detach().numpy()
    out["loss_value"] = tl.detach().numpy()
    out["loss_grad_a"], out["loss_grad_p"], out["loss_grad_n"] = (
        ta.grad.numpy()[:16].copy(), tp.grad.numpy()[:16].copy(), tn.grad.numpy()[:16].copy())
    # train_triplet.py:253,262 restated exactly on the reference's own distances
    allm = (d_n - d_p < 0.1).detach().numpy().flatten()
    out["loss_selected"] = np.where(allm == 1)[0].astype(np.int64)
    out["loss_n_correct"] = np.array(len(np.where(allm == 0)[0]))
    out["loss_mean_diff"] = np.array(np.mean((d_n - d_p).detach().numpy().flatten()))
    # test(): train_triplet.py:348-350 with 8 crops
    out["loss_test_scores"] = d_p.detach().numpy().reshape(N // 8, 8).mean(axis=1)

    # filter on model embeddings (near-collinear at random init: SURVEY section 7)
    d_p2 = pd.forward(ea, ep).detach()
    d_n2 = pd.forward(ea, en).detach()
    out["full_train_d_p"], out["full_train_d_n"] = d_p2.numpy(), d_n2.numpy()
    out["full_train_selected"] = np.where((d_n2 - d_p2 < 0.1).numpy().flatten() == 1)[0].astype(np.int64)

    # CE regime: train_triplet.py:277-287
    logits = torch.from_numpy(rs.randn(12, 16).astype(np.float32))
    labels = torch.from_numpy(rs.randint(0, 16, 12).astype(np.int64))
    out["ce_logits"], out["ce_labels"] = logits.numpy(), labels.numpy()
    out["ce_value"] = torch.nn.CrossEntropyLoss()(logits, labels).numpy()

    # ---------------- eval_metrics.calculate_roc (the importable half of evaluate(); calculate_val raises on
    # scipy >= 1.15) on synthetic verification scores --------------------------------
    import eval_metrics as ref_eval
    rs2 = np.random.RandomState(61)
    n_pairs = 600
    roc_labels = (rs2.rand(n_pairs) < 0.5).astype(np.int64)
    roc_dist = np.where(roc_labels == 1, rs2.normal(6.0, 1.5, n_pairs), rs2.normal(9.0, 1.5, n_pairs)).astype(np.float32)
    tpr, fpr, acc = ref_eval.calculate_roc(np.arange(0, 30, 0.01), roc_dist, roc_labels)
    out["roc_dist"], out["roc_labels"] = roc_dist, roc_labels
    out["roc_tpr_fpr_acc"] = np.array([tpr, fpr, acc], np.float64)

    path = os.path.join(HERE, "reference_outputs.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")
    make_cfg1()


def cfg1_inputs():
    """BASELINE.json configs[1], exactly as bench.py builds it on rank 0: parameters make_state_dict(seed=0,
    num_classes=1211), 256 triplets = anchors | positives | negatives in one [768,1,160,64] randn batch of the
    torch CPU generator seeded 1234."""
    sd = O.make_state_dict(seed=0, num_classes=1211)
    g = torch.Generator(device="cpu").manual_seed(1234)
    return sd, torch.randn(768, 1, 160, 64, generator=g)


def make_cfg1():
    """The unmodified reference at the bench configuration: all 768 embeddings, both distance vectors, the
    triplet loss and the filter's selection (train_triplet.py:251-262 restated on the reference's own distances)."""
    sd, x = cfg1_inputs()
    m = build_ref(sd, 1211).eval()
    with torch.no_grad():
        e = torch.cat([m(x[i:i + 64]) for i in range(0, 768, 64)])
        a, p, n = e[:256], e[256:512], e[512:]
        pd = ref.PairwiseDistance(2)
        d_p, d_n = pd.forward(a, p), pd.forward(a, n)
        loss = ref.TripletMarginLoss(0.1).forward(a, p, n)
    allm = (d_n - d_p < 0.1).numpy().flatten()
    out = {"cfg1_emb": e.numpy(), "cfg1_d_p": d_p.numpy(), "cfg1_d_n": d_n.numpy(), "cfg1_loss": loss.numpy(),
           "cfg1_selected": np.where(allm == 1)[0].astype(np.int64),
           "cfg1_mean_diff": np.array(np.mean((d_n - d_p).numpy().flatten())),
           "cfg1_input_digest": np.array([float(x.double().sum()), float(x.double().abs().sum()), float(x[767, 0, 159, 63])])}
    gap = np.abs(d_n.numpy() - d_p.numpy() - 0.1)
    print("cfg1: selected", len(out["cfg1_selected"]), "of 256; min |d_n - d_p - margin| =", gap.min(),
          "next", np.sort(gap)[1:4])
    path = os.path.join(HERE, "reference_cfg1.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


def make_cfg1_train():
    """The TRAINING step at the bench configuration through the unmodified reference (train_triplet.py:215-223):
    model.train(); out_a, out_p, out_n = model(data_a), model(data_p), model(data_n) on the 3 x 256 utterances of
    configs[1]; TripletMarginLoss(0.1); backward.  Recorded in float32 (what the reference runs) and with the model
    in float64 (the exact derivative of the same function: the tight pin): loss, embeddings, all 36 running
    statistics, a digest of every parameter gradient, and the small gradients (BatchNorm affine, fc bias) in full."""
    sd, x = cfg1_inputs()
    out = {}
    for tag, dt in (("", torch.float32), ("64", torch.float64)):
        m = build_ref(sd, 1211).train().to(dt)
        xs = [x[i * 256:(i + 1) * 256].to(dt) for i in range(3)]
        ea, ep, en = m(xs[0]), m(xs[1]), m(xs[2])
        loss = ref.TripletMarginLoss(0.1).forward(ea, ep, en)
        m.zero_grad()
        loss.backward()
        out[f"cfg1t{tag}_loss"] = loss.detach().numpy()
        out[f"cfg1t{tag}_emb"] = torch.cat([ea, ep, en]).detach().float().numpy()
        for k, v in m.state_dict().items():
            if "running" in k or "num_batches" in k:
                out[f"cfg1t{tag}_stat/" + k] = v.double().numpy() if v.is_floating_point() else v.numpy()
        for k, p in m.named_parameters():
            if p.grad is not None:
                out[f"cfg1t{tag}_grad/" + k] = grad_digest(p.grad)
                if p.grad.numel() <= 2048:
                    out[f"cfg1t{tag}_gfull/" + k] = p.grad.detach().double().numpy()
        print(f"cfg1 train step ({dt}): loss {float(loss):.9f}")
    path = os.path.join(HERE, "reference_cfg1_train.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


# ---------------------------------------------------------------------------------------------------------------
# round 4: off the N(0,1) / random-init distribution, and more than one step deep
# ---------------------------------------------------------------------------------------------------------------
OFFDIST_CASES = {          # name: (torch generator seed, rows, frames, input scale)
    "x15": (4321, 768, 160, 15.0),      # real features are 20*log10 mel energies, std ~ 10-20
    "T100": (4322, 768, 100, 1.0),      # the two ends of BASELINE configs[4]'s length range at the bench batch
    "T800": (4323, 384, 800, 1.0),      # (128 triplets: five times the frames per row)
    "x15_T800": (4324, 96, 800, 15.0),
}


def offdist_inputs(name):
    """the inputs of one off-distribution case, regenerated from its seed (tests rebuild them the same way)"""
    seed, rows, frames, scale = OFFDIST_CASES[name]
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.randn(rows, 1, frames, 64, generator=g) * scale


def make_offdist():
    """The unmodified reference (eval) on the cfg1 parameters with inputs OFF the N(0,1) / 160-frame distribution the
    fp16 path's tolerance and near-tie band were first measured on: embeddings, distances, loss and the filter's
    selection per case (VERDICT r3 'next' #1 (a), (c))."""
    sd = O.make_state_dict(seed=0, num_classes=1211)
    m = build_ref(sd, 1211).eval()
    out = {}
    for name in OFFDIST_CASES:
        x = offdist_inputs(name)
        nt = x.shape[0] // 3
        with torch.no_grad():
            e = torch.cat([m(x[i:i + 32]) for i in range(0, x.shape[0], 32)])
            a, p, n = e[:nt], e[nt:2 * nt], e[2 * nt:]
            pd = ref.PairwiseDistance(2)
            d_p, d_n = pd.forward(a, p), pd.forward(a, n)
            loss = ref.TripletMarginLoss(0.1).forward(a, p, n)
        allm = (d_n - d_p < 0.1).numpy().flatten()
        out[f"{name}_emb"], out[f"{name}_d_p"], out[f"{name}_d_n"] = e.numpy(), d_p.numpy(), d_n.numpy()
        out[f"{name}_loss"] = loss.numpy()
        out[f"{name}_selected"] = np.where(allm == 1)[0].astype(np.int64)
        out[f"{name}_input_digest"] = np.array([float(x.double().sum()), float(x.double().abs().sum()), float(x[-1, 0, -1, -1])])
        gap = np.abs(d_n.numpy() - d_p.numpy() - 0.1)
        print(f"offdist {name}: rows {x.shape[0]}, selected {allm.sum()} of {nt}, min |gap| {gap.min():.3e}, "
              f"mean d_p {float(d_p.mean()):.3f} d_n {float(d_n.mean()):.3f}, |e| max {float(e.abs().max()):.3f}")
    path = os.path.join(HERE, "reference_offdist.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


TRAJ = dict(param_seed=7, corpus_seed=21, speakers=10, utts=24, train_utts=20, frames=160, mix=(0.3, 0.2, 0.9),
            triplets=16, steps=30, lr=0.003, margin=0.1, triplet_seed0=1000)


def trajectory_batches(step):
    """(data_a, data_p, data_n) of training step `step` and the held-out test utterances: shared by the golden maker and
    the GPU test (regenerated from seeds on both sides)"""
    c = O.make_speaker_corpus(TRAJ["corpus_seed"], TRAJ["speakers"], TRAJ["utts"], TRAJ["frames"], mix=TRAJ["mix"])
    a, p, n, _, _ = O.sample_triplets(TRAJ["triplet_seed0"] + step, TRAJ["speakers"], TRAJ["train_utts"], TRAJ["triplets"])
    return [O.gather_utterances(c, i) for i in (a, p, n)]


def trajectory_test_set():
    c = O.make_speaker_corpus(TRAJ["corpus_seed"], TRAJ["speakers"], TRAJ["utts"], TRAJ["frames"], mix=TRAJ["mix"])
    idx = np.array([(s_, u) for s_ in range(TRAJ["speakers"]) for u in range(TRAJ["train_utts"], TRAJ["utts"])], np.int64)
    x = O.gather_utterances(c, idx)
    ii, jj = np.triu_indices(len(idx), 1)
    return x, ii, jj, (idx[ii, 0] == idx[jj, 0])


def make_trajectory():
    """BASELINE configs[3] in miniature (no VoxCeleb here): the triplet-regime loop of train_triplet.py:215-224 through
    the UNMODIFIED reference for 30 steps -- train-mode forwards of a / p / n, TripletMarginLoss, backward, plain SGD (no
    momentum: a smooth trajectory, unlike Adagrad's sign-like first steps) -- on 10 synthetic speakers with learnable
    structure, a fresh batch of 16 triplets per step; then the test-time scoring of train_triplet.py:337-366 on held-out
    utterances of the same speakers (all pairs): distances, the threshold sweep of eval_metrics.calculate_roc, EER."""
    sys.path.insert(0, "/root/reference")
    import eval_metrics as ref_eval
    sd = O.make_state_dict(seed=TRAJ["param_seed"], num_classes=TRAJ["speakers"], randomize_bn=False)
    x_test, ii, jj, same = trajectory_test_set()

    def run(dtype, threads):
        """the 30-step loop + test-time scoring; returns (losses, test embeddings, distances, EER before, EER after, model)"""
        torch.set_num_threads(threads)
        m = build_ref(sd, TRAJ["speakers"]).train().to(dtype)
        opt = torch.optim.SGD(m.parameters(), lr=TRAJ["lr"])

        def eer_now():
            m.eval()
            with torch.no_grad():
                e = m(torch.from_numpy(x_test).to(dtype))
                d = ref.PairwiseDistance(2).forward(e[ii], e[jj]).float().numpy()
            m.train()
            tp, fp = O.roc_sweep(d, same.astype(np.float64), np.arange(0, 30, 0.01))[:2]
            return e.float().numpy(), d, O.equal_error_rate(tp, fp, same.sum(), (~same).sum())

        _, _, eer0 = eer_now()
        losses = []
        for it in range(TRAJ["steps"]):
            xs = [torch.from_numpy(v).to(dtype) for v in trajectory_batches(it)]
            oa, op, on = m(xs[0]), m(xs[1]), m(xs[2])                           # train_triplet.py:215
            loss = ref.TripletMarginLoss(TRAJ["margin"]).forward(oa, op, on)   # :219
            opt.zero_grad()
            loss.backward()
            opt.step()
            losses.append(float(loss))
        e, d, eer = eer_now()
        return np.array(losses, np.float64), e, d, eer0, eer, m

    losses, e, d, eer0, eer, m = run(torch.float32, 8)
    # The loop is CHAOTIC: a hinge or clip mask that rounds the other way changes which triplets / elements carry gradient,
    # and SGD amplifies the difference step after step.  How far the reference is from ITSELF under perturbations of the
    # last bit -- another thread count (another summation order inside ATen / oneDNN), float64 -- is recorded next to the
    # trajectory: it is the yardstick for any other implementation of the same arithmetic.
    losses_alt, _, _, _, eer_alt, _ = run(torch.float32, 3)
    losses64, _, _, _, eer64, _ = run(torch.float64, 8)
    torch.set_num_threads(8)
    tpr, fpr, acc = ref_eval.calculate_roc(np.arange(0, 30, 0.01), d, same.astype(np.float64))
    out = {"traj_loss": losses, "traj_loss_alt_threads": losses_alt, "traj_loss_f64": losses64,
           "traj_test_emb": e, "traj_test_dist": d,
           "traj_test_same": same, "traj_eer": np.array(eer), "traj_eer_before": np.array(eer0),
           "traj_eer_alt_threads": np.array(eer_alt), "traj_eer_f64": np.array(eer64),
           "traj_roc_tpr_fpr_acc": np.array([tpr, fpr, acc], np.float64),
           "traj_final_running_mean_bn1": m.state_dict()["model.bn1.running_mean"].numpy(),
           "traj_final_fc_bias": m.state_dict()["model.fc.bias"].numpy()}
    print("trajectory losses:", " ".join(f"{v:.4f}" for v in losses))
    print("  3 threads, rel:  ", " ".join(f"{abs(a - b) / max(b, 5e-3):.1e}" for a, b in zip(losses_alt, losses)))
    print("  float64, rel:    ", " ".join(f"{abs(a - b) / max(b, 5e-3):.1e}" for a, b in zip(losses64, losses)))
    print(f"mean of first 10 {np.mean(losses[:10]):.4f}, last 10 {np.mean(losses[-10:]):.4f}; EER {eer0:.4f} -> {eer:.4f} "
          f"(3 threads {eer_alt:.4f}, float64 {eer64:.4f})")
    path = os.path.join(HERE, "reference_trajectory.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "cfg1_train":
        make_cfg1_train()
    elif len(sys.argv) > 1 and sys.argv[1] == "offdist":
        make_offdist()
    elif len(sys.argv) > 1 and sys.argv[1] == "trajectory":
        make_trajectory()
    else:
        main()
        make_cfg1_train()
        make_offdist()
        make_trajectory()
