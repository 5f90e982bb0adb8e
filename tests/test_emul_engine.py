"""Host orchestration (Engine) + unmodified kernels on the host emulator vs the numpy oracle.
CPU tensors are handed to the emulated library here ONLY to test host logic and kernel indexing
without a GPU; the product module refuses non-CUDA tensors."""
import numpy as np
import pytest
import torch

import deepspeaker_oracle as O
from conftest import rel_err
from emul_util import emul_lib
from deepspeaker_pytorch_amd.engine import BNParams, Engine


def torch_sd(sd):
    return {k: torch.from_numpy(np.array(v)) for k, v in sd.items()}


def bn_names(n_stages):
    out = []
    for i in range(1, n_stages + 1):
        out += [f"model.bn{i}", f"model.layer{i}.0.bn1", f"model.layer{i}.0.bn2"]
    return out


def make_bns(tsd, n_stages):
    return {n: BNParams(tsd[n + ".weight"], tsd[n + ".bias"], tsd[n + ".running_mean"], tsd[n + ".running_var"])
            for n in bn_names(n_stages)}


@pytest.mark.parametrize("n_stages,B,T", [(4, 2, 32), (2, 3, 21), (4, 1, 50)])
def test_forward_eval(n_stages, B, T):
    eng = Engine(emul_lib())
    sd = O.make_state_dict(seed=5 + n_stages, num_classes=4, n_stages=n_stages)
    x = O.make_input(seed=6, batch=B, frames=T)
    tsd = torch_sd(sd)
    pw = eng.pack_weights(tsd, n_stages)
    folded = {n: eng.bn_fold(b) for n, b in make_bns(tsd, n_stages).items()}
    taps = {}
    e = eng.forward_eval(torch.from_numpy(x), pw, folded, taps)
    otaps = {}
    ref = O.forward(sd, x, n_stages=n_stages, dtype=np.float64, taps=otaps)
    for k, v in taps.items():
        got = v.numpy().transpose(0, 3, 1, 2)
        assert rel_err(got, otaps[k]) < 1e-5, k
    assert rel_err(e.numpy(), ref) < 1e-5


def test_forward_train_matches_oracle_and_updates_running_stats():
    eng = Engine(emul_lib())
    n_stages, B, T = 4, 2, 32
    sd = O.make_state_dict(seed=9, num_classes=4)
    x = O.make_input(seed=10, batch=B, frames=T)
    tsd = torch_sd(sd)
    pw = eng.pack_weights(tsd, n_stages)
    bns = make_bns(tsd, n_stages)
    e, saved = eng.forward_train(torch.from_numpy(x), pw, bns)
    new = {}
    ref = O.forward(sd, x, train=True, dtype=np.float64, new_stats=new)
    assert rel_err(e.numpy(), ref) < 2e-5
    for n in bn_names(n_stages):
        assert rel_err(bns[n].running_mean.numpy(), new[n + ".running_mean"]) < 1e-5, n
        assert rel_err(bns[n].running_var.numpy(), new[n + ".running_var"]) < 1e-5, n
    assert set(saved.raws) == set(bn_names(n_stages))


def test_loss_side(golden):
    eng = Engine(emul_lib())
    rs = np.random.RandomState(41)
    N = 96
    base = rs.randn(N, 512).astype(np.float32)
    a = (base / np.linalg.norm(base, axis=1, keepdims=True) * 10).astype(np.float32)
    p = a + rs.randn(N, 512).astype(np.float32) * 0.05
    n = a + rs.randn(N, 512).astype(np.float32) * 0.05
    ta, tp, tn = (torch.from_numpy(v) for v in (a, p, n))
    loss, d_p, d_n = eng.triplet_margin(ta, tp, tn, 0.1)
    assert rel_err(d_p.numpy(), golden["loss_d_p"]) < 1e-6
    assert rel_err(d_n.numpy(), golden["loss_d_n"]) < 1e-6
    assert abs(float(loss) - float(golden["loss_value"])) < 1e-6
    idx, count, mean_diff = eng.triplet_filter(d_p, d_n, 0.1)
    k = int(count)
    np.testing.assert_array_equal(idx.numpy()[:k], golden["loss_selected"])
    assert abs(float(mean_diff) - float(golden["loss_mean_diff"])) < 1e-6
    assert rel_err(eng.pairwise_distance(ta, tp).numpy(), golden["loss_d_p"]) < 1e-6


@pytest.mark.parametrize("N,cap", [(1, 4), (5, 2), (256, 8), (700, 8)])
def test_triplet_tail_scan_and_refinement(N, cap):
    """ds_triplet_tail_f32 (loss + ordered filter + mean difference + near-tie list in one scan) against the
    oracle; then the near-tie refinement: patched distances and a re-scan give the selection of the patched set."""
    eng = Engine(emul_lib())
    rs = np.random.RandomState(100 + N)
    D = 64
    a = rs.randn(N, D).astype(np.float32)
    p = a + rs.randn(N, D).astype(np.float32) * 0.05
    n = a + rs.randn(N, D).astype(np.float32) * 0.05
    margin, band = 0.1, 0.02
    t = eng.triplet_tail(*(torch.from_numpy(v) for v in (a, p, n)), margin, band=band, amb_cap=cap)
    ref_loss, d_p, d_n = O.triplet_margin_loss(a, p, n, margin)
    ref_idx, _, md = O.triplet_filter(d_p, d_n, margin)
    assert rel_err(t["d_p"].numpy(), d_p) < 1e-6 and rel_err(t["d_n"].numpy(), d_n) < 1e-6
    assert abs(float(t["loss"]) - float(ref_loss)) < 1e-6
    assert int(t["count"]) == len(ref_idx)
    np.testing.assert_array_equal(t["idx"].numpy()[:len(ref_idx)], ref_idx)
    assert abs(float(t["mean_diff"]) - md) < 1e-6
    amb_ref = np.where(np.abs(t["d_n"].numpy() - t["d_p"].numpy() - np.float32(margin)) < np.float32(band))[0]
    assert int(t["amb_count"]) == len(amb_ref)
    k = min(cap, len(amb_ref))
    np.testing.assert_array_equal(t["amb_idx"].numpy()[:k], amb_ref[:k])
    assert (t["amb_idx"].numpy()[k:] == 0).all()
    # a buffer rewritten IN PLACE through its raw pointer (as a HIP graph's static output or a collective's
    # destination is: same address, same torch version counter) must give the new contents' result
    ta, tp, tn = (torch.from_numpy(v.copy()) for v in (a, p, n))
    t1 = eng.triplet_tail(ta, tp, tn, margin)
    ta.numpy()[:] = n                  # numpy view: bypasses torch's version counter
    t2 = eng.triplet_tail(ta, tp, tn, margin)
    ref_loss2, _, _ = O.triplet_margin_loss(n, p, n, margin)
    assert abs(float(t2["loss"]) - float(ref_loss2)) < 1e-6 and abs(float(t1["loss"]) - float(ref_loss)) < 1e-6
    # refinement: "re-embedded" rows for the cap slots (anchors | positives | negatives)
    e_ref = rs.randn(3 * cap, D).astype(np.float32)
    dp2, dn2 = t["d_p"].clone(), t["d_n"].clone()
    eng.lib.call("ds_refine_distances_f32", eng._p(torch.from_numpy(e_ref)), eng._p(t["amb_idx"]), eng._p(t["amb_count"]),
                 cap, eng._p(dp2), eng._p(dn2), D, None)
    exp_p, exp_n = t["d_p"].numpy().copy(), t["d_n"].numpy().copy()
    for s_ in range(k):
        i = amb_ref[s_]
        exp_p[i] = O.pairwise_distance(e_ref[s_:s_ + 1], e_ref[cap + s_:cap + s_ + 1])[0]
        exp_n[i] = O.pairwise_distance(e_ref[s_:s_ + 1], e_ref[2 * cap + s_:2 * cap + s_ + 1])[0]
    assert rel_err(dp2.numpy(), exp_p) < 1e-6 and rel_err(dn2.numpy(), exp_n) < 1e-6
    idx, count = torch.empty(N, dtype=torch.int64), torch.empty(1, dtype=torch.int32)
    md2, loss2 = torch.empty(1), torch.empty(1)
    eng.lib.call("ds_triplet_scan_f32", eng._p(dp2), eng._p(dn2), margin, eng._p(loss2), eng._p(idx), eng._p(count),
                 eng._p(md2), N, None)
    ref_idx2, _, _ = O.triplet_filter(dp2.numpy(), dn2.numpy(), margin)
    np.testing.assert_array_equal(idx.numpy()[:int(count)], ref_idx2)
    assert abs(float(loss2) - float(np.maximum(margin + dp2.numpy() - dn2.numpy(), 0).mean())) < 1e-6


@pytest.mark.parametrize("N,cap,base", [(5, 8, 3), (256, 8, 250), (700, 16, 0), (40, 4, 7)])
def test_triplet_tail_probe_slots_and_error_readout(N, cap, base):
    """ds_triplet_tail_probe_f32: the slots near ties leave unused hold the probe triplets (base + k) mod N;
    ds_refine_distances_probe_f32 patches EVERY slot and reports the largest |change of d_n - d_p| over them (read from the
    unpatched distances, so a probe that repeats a near tie's index cannot see a half-patched pair) and the slot count."""
    eng = Engine(emul_lib())
    rs = np.random.RandomState(300 + N)
    D = 64
    a = rs.randn(N, D).astype(np.float32)
    p = a + rs.randn(N, D).astype(np.float32) * 0.05
    n = a + rs.randn(N, D).astype(np.float32) * 0.05
    margin, band = 0.1, 0.01
    t = eng.triplet_tail(*(torch.from_numpy(v) for v in (a, p, n)), margin, band=band, amb_cap=cap, probe_base=base)
    t0 = eng.triplet_tail(*(torch.from_numpy(v) for v in (a, p, n)), margin, band=band, amb_cap=cap)
    for k_ in ("d_p", "d_n", "loss", "count", "mean_diff", "amb_count"):        # probes change nothing but the unused slots
        assert torch.equal(t[k_], t0[k_]), k_
    amb_ref = np.where(np.abs(t["d_n"].numpy() - t["d_p"].numpy() - np.float32(margin)) < np.float32(band))[0]
    k = min(cap, len(amb_ref))
    slots = t["amb_idx"].numpy()
    np.testing.assert_array_equal(slots[:k], amb_ref[:k])
    np.testing.assert_array_equal(slots[k:], (base + np.arange(cap - k)) % N)
    e_ref = rs.randn(3 * cap, D).astype(np.float32)
    dp2, dn2 = t["d_p"].clone(), t["d_n"].clone()
    err = torch.full((4,), -1.0)
    ta_, tp_, tn_ = (torch.from_numpy(v) for v in (a, p, n))
    te_ref = torch.from_numpy(e_ref)
    eng.lib.call("ds_refine_distances_probe_f32", eng._p(te_ref), eng._p(t["amb_idx"]),
                 eng._p(t["amb_count"]), cap, eng._p(dp2), eng._p(dn2), eng._p(t["d_p"]), eng._p(t["d_n"]), eng._p(ta_), eng._p(tp_),
                 eng._p(tn_), D, eng._p(err), None)
    exp_p, exp_n = t["d_p"].numpy().copy(), t["d_n"].numpy().copy()
    worst = 0.0
    for s_ in range(cap):
        i = slots[s_]
        new_p = O.pairwise_distance(e_ref[s_:s_ + 1], e_ref[cap + s_:cap + s_ + 1])[0]
        new_n = O.pairwise_distance(e_ref[s_:s_ + 1], e_ref[2 * cap + s_:2 * cap + s_ + 1])[0]
        worst = max(worst, abs(float(np.float32(new_n) - np.float32(new_p)) - float(t["d_n"][i] - t["d_p"][i])))
        exp_p[i], exp_n[i] = new_p, new_n           # (repeated indices: the last slot's values; all slots agree below)
    assert int(err[1]) == cap and abs(float(err[0]) - worst) < 1e-5 * max(1.0, worst)
    # the embedding comparison on the sampled rows: max |e_ref - emb| and max |e_ref| over anchors | positives | negatives
    ediff = max(np.abs(e_ref[k_ * cap + s_] - src[slots[s_]]).max() for s_ in range(cap) for k_, src in enumerate((a, p, n)))
    assert abs(float(err[2]) - ediff) < 1e-6 * max(1.0, ediff) and abs(float(err[3]) - np.abs(e_ref).max()) < 1e-6
    err_no = torch.full((4,), -1.0)         # without the path's embeddings: nothing compared
    dpx, dnx = t["d_p"].clone(), t["d_n"].clone()
    eng.lib.call("ds_refine_distances_probe_f32", eng._p(te_ref), eng._p(t["amb_idx"]), eng._p(t["amb_count"]), cap, eng._p(dpx),
                 eng._p(dnx), eng._p(t["d_p"]), eng._p(t["d_n"]), None, None, None, D, eng._p(err_no), None)
    assert float(err_no[2]) == 0.0 and float(err_no[0]) == float(err[0])
    untouched = np.setdiff1d(np.arange(N), slots)
    np.testing.assert_array_equal(dp2.numpy()[untouched], t["d_p"].numpy()[untouched])
    if len(set(slots.tolist())) == cap:                         # no index twice: every patched value is determined
        assert rel_err(dp2.numpy(), exp_p) < 1e-6 and rel_err(dn2.numpy(), exp_n) < 1e-6
    # the same buffers as "before" and patched are refused
    rc = eng.lib.raw("ds_refine_distances_probe_f32")(eng._p(te_ref), eng._p(t["amb_idx"]), eng._p(t["amb_count"]),
                                                      cap, eng._p(dp2), eng._p(dn2), eng._p(dp2), eng._p(dn2), None, None, None, D,
                                                      eng._p(err), None)
    assert rc != 0
    # round 6: ds_refine_distances_fused_f32 = two clones + the call above + the count's read-back in ONE launch: the
    # outputs are WRITTEN (garbage in), bitwise the patched clones; err5[:4] the same read-out, err5[4] the near-tie count
    dp3, dn3 = torch.full((N,), float("nan")), torch.full((N,), float("nan"))
    err5 = torch.full((5,), -1.0)
    eng.lib.call("ds_refine_distances_fused_f32", eng._p(te_ref), eng._p(t["amb_idx"]), eng._p(t["amb_count"]), cap, eng._p(dp3),
                 eng._p(dn3), eng._p(t["d_p"]), eng._p(t["d_n"]), eng._p(ta_), eng._p(tp_), eng._p(tn_), N, D, eng._p(err5), None)
    assert torch.equal(dp3, dp2) and torch.equal(dn3, dn2) and torch.equal(err5[:4], err)
    assert int(err5[4]) == int(t["amb_count"])


@pytest.mark.parametrize("N", [1, 5, 256, 700])
def test_filter_sizes(N):
    eng = Engine(emul_lib())
    rs = np.random.RandomState(N)
    d_p = torch.from_numpy(rs.rand(N).astype(np.float32))
    d_n = torch.from_numpy(rs.rand(N).astype(np.float32))
    idx, count, mean_diff = eng.triplet_filter(d_p, d_n, 0.1)
    ref, _, md = O.triplet_filter(d_p.numpy(), d_n.numpy(), 0.1)
    assert int(count) == len(ref)
    np.testing.assert_array_equal(idx.numpy()[:len(ref)], ref)
    assert abs(float(mean_diff) - md) < 1e-6


@pytest.mark.parametrize("n_stages,B,T,seed", [(4, 2, 32, 16), (2, 2, 23, 13)])
def test_backward_matches_oracle(n_stages, B, T, seed):
    """Whole train-mode backward (orchestration in backward.py + the unmodified kernels) vs the numpy
    restatement that is pinned to the reference evaluated in float64.

    Seeds are chosen so that no activation sits within fp32 rounding of a clip edge: with so few
    pixels ONE flipped clipped-ReLU mask between an fp32 and an fp64 forward moves early-layer
    gradients by ~1e-2 (seed 13 at 4 stages does, for the numpy fp32 restatement and for the kernels
    alike; seed 14 has a stage-1 pre-activation 1e-7 from zero whose sign depends on whether z * scale + shift
    is one fused multiply-add, as on the GPU and in ds_bn_affine, or two roundings) -- a property of the
    function, not of the implementation."""
    from deepspeaker_pytorch_amd.backward import backward_train
    eng = Engine(emul_lib())
    sd = O.make_state_dict(seed=seed, num_classes=4, n_stages=n_stages)
    x = O.make_input(seed=seed + 1, batch=B, frames=T)
    tsd = torch_sd(sd)
    pw = eng.pack_weights(tsd, n_stages, with_dgrad=True)
    bns = make_bns(tsd, n_stages)
    e, saved = eng.forward_train(torch.from_numpy(x), pw, bns)
    ge = np.random.RandomState(3).randn(B, 512).astype(np.float32)
    grads = backward_train(eng, {n: b.weight for n, b in bns.items()}, pw, saved, torch.from_numpy(ge))
    cache = {}
    O.forward(sd, x, train=True, n_stages=n_stages, dtype=np.float64, cache=cache)
    ref = O.backward(sd, cache, x, ge, n_stages=n_stages)
    assert set(grads) == set(ref)
    for k, v in ref.items():
        got = grads[k].numpy()
        assert got.shape == v.shape, k
        err = np.linalg.norm(got - v) / max(np.linalg.norm(v), 1e-30)
        assert err < 5e-5, (k, err)


@pytest.mark.parametrize("precision,tol", [("bf16x3", 3e-5), ("bf16", 3e-2), ("f16", 1e-3)])
def test_forward_eval_bf16_precisions(precision, tol):
    eng = Engine(emul_lib())
    n_stages, B, T = 4, 2, 32
    sd = O.make_state_dict(seed=9, num_classes=4, n_stages=n_stages)
    x = O.make_input(seed=6, batch=B, frames=T)
    tsd = torch_sd(sd)
    pw = eng.pack_weights(tsd, n_stages, with_bf16=precision != "f16", with_f16=precision == "f16")
    folded = {n: eng.bn_fold(b) for n, b in make_bns(tsd, n_stages).items()}
    e = eng.forward_eval(torch.from_numpy(x), pw, folded, precision=precision)
    ref = O.forward(sd, x, n_stages=n_stages, dtype=np.float64)
    err = rel_err(e.numpy(), ref)
    print(precision, err)
    assert err < tol


@pytest.mark.parametrize("kind", ["adagrad", "sgd", "sgd_plain", "adam"])
def test_fused_optimizers_match_torch(kind):
    """One-launch multi-tensor steps vs torch.optim with the reference's hyper-parameters
    (train_triplet.py:369-383); the checker here is torch's own CPU optimizer."""
    from deepspeaker_pytorch_amd import optim as fo
    eng = Engine(emul_lib())
    rs = np.random.RandomState(3)
    shapes = [(64, 1, 5, 5), (17,), (33000,), (128, 64, 3, 3)]
    ref_p = [torch.nn.Parameter(torch.from_numpy(rs.randn(*s).astype(np.float32))) for s in shapes]
    our_p = [torch.nn.Parameter(p.detach().clone()) for p in ref_p]
    if kind == "adagrad":
        ref = torch.optim.Adagrad(ref_p, lr=0.1, lr_decay=1e-4, weight_decay=1e-3)
        ours = fo.FusedAdagrad(our_p, lr=0.1, lr_decay=1e-4, weight_decay=1e-3)
    elif kind == "sgd":
        ref = torch.optim.SGD(ref_p, lr=0.1, momentum=0.9, dampening=0.9, weight_decay=1e-3)
        ours = fo.FusedSGD(our_p, lr=0.1, momentum=0.9, dampening=0.9, weight_decay=1e-3)
    elif kind == "sgd_plain":       # no momentum: the optimizer has no state tensors at all (a null state table)
        ref = torch.optim.SGD(ref_p, lr=0.05)
        ours = fo.FusedSGD(our_p, lr=0.05)
    else:
        ref = torch.optim.Adam(ref_p, lr=0.01, weight_decay=1e-3)
        ours = fo.FusedAdam(our_p, lr=0.01, weight_decay=1e-3)
    ours._engine = eng
    for it in range(4):
        for k, (a, b) in enumerate(zip(ref_p, our_p)):
            if k == 1 and it < 2:       # a parameter that starts receiving gradients later (the classifier head
                a.grad = b.grad = None  # when the regime switches): its own step count / fresh momentum buffer
                continue
            g = torch.from_numpy(rs.randn(*a.shape).astype(np.float32))
            a.grad, b.grad = g.clone(), g.clone()
        versions = [b._version for b in our_p]
        ref.step()
        ours.step()
        for k, (a, b) in enumerate(zip(ref_p, our_p)):
            assert rel_err(b.detach().numpy(), a.detach().numpy()) < 2e-6, (kind, it, k)
            # raw-pointer updates must be visible to version-keyed caches (DeepSpeakerModel._packed / _folded)
            assert (b._version > versions[k]) == (b.grad is not None), (kind, it, k)
    # state interchange: our state_dict loads into torch's optimizer and vice versa
    sd = ours.state_dict()
    if kind != "sgd_plain":
        assert set(sd["state"][0].keys()) == set(ref.state_dict()["state"][0].keys())
    ref.load_state_dict(sd)


@pytest.mark.parametrize("kind", ["adagrad", "sgd", "adam"])
def test_fused_optimizers_device_side_step_count(kind):
    """enable_device_step() (what train_graph.GraphedTripletStep uses: a captured step must replay with the step count of
    the replay): Adagrad's decayed learning rate, Adam's bias corrections and SGD's "first step" come from an int32 counter
    on the device that each step() bumps there.  Same trajectories as torch.optim over five steps; a step skipped by the
    overflow flag neither updates nor counts; sync_host_steps() brings state[p]['step'] up to date."""
    from deepspeaker_pytorch_amd import optim as fo
    eng = Engine(emul_lib())
    rs = np.random.RandomState(5)
    shapes = [(64, 1, 5, 5), (17,), (20000,)]
    ref_p = [torch.nn.Parameter(torch.from_numpy(rs.randn(*s).astype(np.float32))) for s in shapes]
    our_p = [torch.nn.Parameter(p.detach().clone()) for p in ref_p]
    if kind == "adagrad":
        ref, ours = torch.optim.Adagrad(ref_p, lr=0.1, lr_decay=1e-2), fo.FusedAdagrad(our_p, lr=0.1, lr_decay=1e-2)
    elif kind == "sgd":
        ref = torch.optim.SGD(ref_p, lr=0.1, momentum=0.9, dampening=0.9)
        ours = fo.FusedSGD(our_p, lr=0.1, momentum=0.9, dampening=0.9)
    else:
        ref, ours = torch.optim.Adam(ref_p, lr=0.01), fo.FusedAdam(our_p, lr=0.01)
    ours._engine = eng
    ours.enable_device_step()
    flag = torch.zeros(1, dtype=torch.int32)
    ours.skip_flag = flag
    for it in range(6):
        for a, b in zip(ref_p, our_p):
            g = torch.from_numpy(rs.randn(*a.shape).astype(np.float32))
            a.grad, b.grad = g.clone(), g.clone()
        if it == 2:                             # an overflowed step: skipped on the device, and NOT counted there
            flag.fill_(1)
            before = [b.detach().clone() for b in our_p]
            ours.step()
            assert all(torch.equal(b.detach(), k) for b, k in zip(our_p, before)) and int(ours._dev_step) == 2 and int(flag) == 0
            continue
        ref.step()
        ours.step()
        for k, (a, b) in enumerate(zip(ref_p, our_p)):
            assert rel_err(b.detach().numpy(), a.detach().numpy()) < 2e-6, (kind, it, k)
    assert int(ours._dev_step) == 5
    ours.sync_host_steps()
    if kind != "sgd":
        assert all(float(ours.state[p]["step"]) == 5.0 for p in our_p)


def test_scoring_and_roc(golden):
    from deepspeaker_pytorch_amd import scoring
    scoring._engine_override = Engine(emul_lib())
    try:
        d = torch.from_numpy(golden["roc_dist"])
        lab = torch.from_numpy(golden["roc_labels"])
        v = scoring.evaluate(d, lab)
        np.testing.assert_allclose([v.tpr, v.fpr, v.accuracy], golden["roc_tpr_fpr_acc"], atol=1e-6)
        thr = np.arange(0, 30, 0.01)
        tp, fp, best, *_ = O.roc_sweep(golden["roc_dist"], golden["roc_labels"], thr)
        np.testing.assert_array_equal(v.tp.numpy(), tp)
        np.testing.assert_array_equal(v.fp.numpy(), fp)
        assert abs(v.threshold - thr[best]) < 1e-5
        issame = golden["roc_labels"].astype(bool)
        assert abs(v.eer - O.equal_error_rate(tp, fp, issame.sum(), (~issame).sum())) < 1e-5
        # test-time score: mean of 8 crop-pair distances (train_triplet.py:348-350)
        rs = np.random.RandomState(2)
        a, p = rs.randn(40, 512).astype(np.float32), rs.randn(40, 512).astype(np.float32)
        s = scoring.trial_scores(torch.from_numpy(a), torch.from_numpy(p), 8)
        assert rel_err(s.numpy(), O.test_scores(a, p, 8)) < 1e-6
    finally:
        scoring._engine_override = None


def test_feature_store_crops():
    from deepspeaker_pytorch_amd import data
    data._engine_override = Engine(emul_lib())
    try:
        rs = np.random.RandomState(4)
        utts = [rs.randn(t, 64).astype(np.float32) for t in (50, 33, 120)]
        fs = data.FeatureStore(utts, device="cpu")
        idx, st = [2, 0, 1, 2], [10, 0, 20, 100]
        x = fs.crops(idx, st, 32).numpy()
        assert x.shape == (4, 1, 32, 64)
        for b, (u, s) in enumerate(zip(idx, st)):
            ref = np.zeros((32, 64), np.float32)
            seg = utts[u][s:s + 32]
            ref[:len(seg)] = seg                               # zero padded past the utterance end
            np.testing.assert_array_equal(x[b, 0], ref)
        a, p, n = fs.triplets([0, 1], [0, 1], [2, 2], [[1, 2], [3, 0], [5, 60]], 16)
        np.testing.assert_array_equal(n[1, 0].numpy(), utts[2][60:76])
        with pytest.raises(IndexError):
            fs.crops([3], [0], 8)
    finally:
        data._engine_override = None


def test_classifier_head_and_cross_entropy(golden):
    """forward_classifier's GEMM + the CE of train_triplet.py:277-287, forward and backward, vs torch on CPU."""
    from deepspeaker_pytorch_amd.model import HeadPack, _CrossEntropyFn, _HeadCrossEntropyFn, _LinearHeadFn
    eng = Engine(emul_lib())
    rs = np.random.RandomState(12)
    M, K, N = 12, 512, 21
    x = torch.from_numpy(rs.randn(M, K).astype(np.float32)).requires_grad_(True)
    w = torch.from_numpy((rs.randn(N, K) * 0.05).astype(np.float32)).requires_grad_(True)
    b = torch.from_numpy(rs.randn(N).astype(np.float32)).requires_grad_(True)
    labels = torch.from_numpy(rs.randint(0, N, M).astype(np.int64))
    pack = HeadPack(eng, w, b)
    # the fused GEMM + log-softmax + NLL (row max / lse in the reduction epilogue) must equal the two-step form
    xf, wf, bf = (t.detach().clone().requires_grad_(True) for t in (x, w, b))
    fused = _HeadCrossEntropyFn.apply(xf, wf, bf, labels, eng, pack)
    fused.backward()
    logits = _LinearHeadFn.apply(x, w, b, eng, pack)
    loss = _CrossEntropyFn.apply(logits, labels, eng)
    loss.backward()
    assert abs(float(fused) - float(loss)) < 1e-6
    for a, r in ((xf, x), (wf, w), (bf, b)):
        assert rel_err(a.grad.numpy(), r.grad.numpy()) < 1e-6
    xr, wr, br = (t.detach().clone().requires_grad_(True) for t in (x, w, b))
    lr = torch.nn.functional.linear(xr, wr, br)
    ref = torch.nn.CrossEntropyLoss()(lr, labels)
    ref.backward()
    assert rel_err(logits.detach().numpy(), lr.detach().numpy()) < 1e-5
    assert abs(float(loss) - float(ref)) < 1e-5
    for a, r in ((x, xr), (w, wr), (b, br)):
        assert rel_err(a.grad.numpy(), r.grad.numpy()) < 1e-5
    # the reference-recorded CE value
    l2 = _CrossEntropyFn.apply(torch.from_numpy(golden["ce_logits"]), torch.from_numpy(golden["ce_labels"]), eng)
    assert abs(float(l2) - float(golden["ce_value"])) < 1e-6


@pytest.mark.parametrize("precision", ["f32", "bf16x3", "f16"])
def test_planned_eval_forward_equals_unplanned(precision):
    """The cached launch plan must be the same computation as the step-by-step forward, call after call
    (buffers are re-used) and across shapes."""
    eng = Engine(emul_lib())
    n_stages = 2
    sd = O.make_state_dict(seed=19, num_classes=4, n_stages=n_stages)
    tsd = torch_sd(sd)
    pw = eng.pack_weights(tsd, n_stages, with_bf16=True, with_f16=True)
    folded = {n: eng.bn_fold(b) for n, b in make_bns(tsd, n_stages).items()}
    for seed, (B, T) in enumerate([(2, 24), (3, 17), (2, 24)]):
        x = torch.from_numpy(O.make_input(seed=seed, batch=B, frames=T))
        e1 = eng.forward_eval(x, pw, folded, precision=precision)
        e2 = eng.forward_eval_planned(x, pw, folded, precision=precision)
        e3 = eng.forward_eval_planned(x, pw, folded, precision=precision)
        assert torch.equal(e1, e2) and torch.equal(e2, e3)
    # drop_eval_plans: the cache is emptied (its activation buffers go back to the allocator) and the next call plans afresh
    assert len(eng._eval_plans) == 2
    assert eng.drop_eval_plans() > 0 and len(eng._eval_plans) == 0 and eng.drop_eval_plans() == 0
    assert torch.equal(eng.forward_eval_planned(x, pw, folded, precision=precision), e1) and len(eng._eval_plans) == 1


def test_training_step_bf16x3_matches_oracle():
    """Train-mode forward + backward with the bf16x3 convolutions (forward, data and filter gradients)."""
    from deepspeaker_pytorch_amd.backward import backward_train
    eng = Engine(emul_lib())
    n_stages, B, T, seed = 2, 2, 23, 13
    sd = O.make_state_dict(seed=seed, num_classes=4, n_stages=n_stages)
    x = O.make_input(seed=seed + 1, batch=B, frames=T)
    tsd = torch_sd(sd)
    pw = eng.pack_weights(tsd, n_stages, with_dgrad=True, with_bf16=True)
    bns = make_bns(tsd, n_stages)
    e, saved = eng.forward_train(torch.from_numpy(x), pw, bns, precision="bf16x3")
    ge = np.random.RandomState(3).randn(B, 512).astype(np.float32)
    grads = backward_train(eng, {n: b.weight for n, b in bns.items()}, pw, saved, torch.from_numpy(ge),
                           precision="bf16x3")
    cache = {}
    ref_e = O.forward(sd, x, train=True, n_stages=n_stages, dtype=np.float64, cache=cache)
    assert rel_err(e.numpy(), ref_e) < 5e-5
    ref = O.backward(sd, cache, x, ge, n_stages=n_stages)
    for k, v in ref.items():
        err = np.linalg.norm(grads[k].numpy() - v) / max(np.linalg.norm(v), 1e-30)
        assert err < 3e-4, (k, err)


@pytest.mark.parametrize("precision", ["f32", "bf16x3"])
def test_grouped_triplet_forward_backward_equals_three_calls(precision):
    """Engine.forward_train_group + one backward over the concatenated batch == the reference's call pattern
    (three train-mode forwards, three backward passes accumulated): embeddings and saved statistics bitwise,
    running statistics bitwise (three sequential momentum updates), gradients to summation-order rounding."""
    from deepspeaker_pytorch_amd.backward import backward_train
    eng = Engine(emul_lib())
    n_stages, B, T = 2, 2, 23
    sd = O.make_state_dict(seed=23, num_classes=4, n_stages=n_stages)
    xs = [torch.from_numpy(O.make_input(seed=30 + i, batch=B, frames=T)) for i in range(3)]
    ges = [torch.from_numpy(np.random.RandomState(40 + i).randn(B, 512).astype(np.float32)) for i in range(3)]
    x3 = precision == "bf16x3"

    def fresh():
        tsd = torch_sd(sd)
        tsd = {k: v.clone() for k, v in tsd.items()}
        return tsd, eng.pack_weights(tsd, n_stages, with_dgrad=True, with_bf16=x3), make_bns(tsd, n_stages)

    tsd1, pw1, bns1 = fresh()
    bn_w1 = {n: tsd1[n + ".weight"] for n in bn_names(n_stages)}
    sep_e, sep_g = [], {}
    for x, ge in zip(xs, ges):
        e, saved = eng.forward_train(x, pw1, bns1, precision=precision)
        sep_e.append(e)
        for k, v in backward_train(eng, bn_w1, pw1, saved, ge, precision=precision).items():
            sep_g[k] = sep_g[k] + v if k in sep_g else v
    tsd2, pw2, bns2 = fresh()
    bn_w2 = {n: tsd2[n + ".weight"] for n in bn_names(n_stages)}
    embs, saved = eng.forward_train_group(xs, pw2, bns2, precision=precision)
    for a, b in zip(sep_e, embs):
        assert torch.equal(a, b)
    for n in bn_names(n_stages):
        assert torch.equal(bns1[n].running_mean, bns2[n].running_mean), n
        assert torch.equal(bns1[n].running_var, bns2[n].running_var), n
        assert len(saved.stats[n]) == 3
    grads = backward_train(eng, bn_w2, pw2, saved, torch.cat(ges), precision=precision)
    assert set(grads) == set(sep_g)
    for k, v in sep_g.items():
        err = float((grads[k] - v).norm() / v.norm().clamp_min(1e-30))
        assert err < 2e-6, (k, err)


@pytest.mark.parametrize("grouped", [False, True])
def test_fused_data_gradient_and_batchnorm_backward_equals_two_steps(grouped):
    """backward._dgrad_bn_bwd (ds_conv_dgrad_bnbwd_bf16: the 3x3 data gradient with the BatchNorm-backward reduction and
    the clipped-ReLU mask, re-derived from the pre-activation, in its epilogue) against the two-step sequence it
    replaces (ds_conv_dgrad_bf16, then ds_bn_bwd[_group]_f32 reading the stored activation for the mask): the fused
    entry point is taken for 2 layers per stage, and every gradient agrees to summation-order rounding."""
    from deepspeaker_pytorch_amd import backward
    eng = Engine(emul_lib())
    n_stages, B, T = 2, 2, 23
    sd = O.make_state_dict(seed=23, num_classes=4, n_stages=n_stages)
    xs = [torch.from_numpy(O.make_input(seed=30 + i, batch=B, frames=T)) for i in range(3 if grouped else 1)]
    ge = torch.from_numpy(np.random.RandomState(40).randn(B * len(xs), 512).astype(np.float32))
    tsd = torch_sd(sd)
    pw = eng.pack_weights(tsd, n_stages, with_dgrad=True, with_bf16=True)
    bns = make_bns(tsd, n_stages)
    bn_w = {n: tsd[n + ".weight"] for n in bn_names(n_stages)}
    if grouped:
        _, saved = eng.forward_train_group(xs, pw, bns, precision="bf16x3")
    else:
        _, saved = eng.forward_train(xs[0], pw, bns, precision="bf16x3")
    out = {}
    try:
        for fuse in (True, False):
            backward.FUSE_DGRAD_BN_BWD = fuse
            eng.lib.trace = {}
            out[fuse] = backward.backward_train(eng, bn_w, pw, saved, ge, precision="bf16x3")
            assert eng.lib.trace.get("ds_conv_dgrad_bnbwd_bf16", 0) == (2 * n_stages if fuse else 0)
            assert eng.lib.trace.get("ds_conv_dgrad_s2_bnbwd_bf16", 0) == (n_stages - 1 if fuse else 0)
            assert eng.lib.trace.get("ds_bn_bwd_group_finish_f32", 0) == (3 * n_stages - 1 if fuse else 0)
    finally:
        backward.FUSE_DGRAD_BN_BWD = True
        eng.lib.trace = None
    for k, v in out[False].items():
        err = float((out[True][k] - v).norm() / v.norm().clamp_min(1e-30))
        assert err < 1e-5, (k, err)          # the two reductions add the same terms in a different order


@pytest.mark.parametrize("precision", ["f32", "f16"])
def test_masked_variable_length_batch_is_bit_identical_to_single_forwards(precision):
    """BASELINE configs[4]: utterances of different lengths in one zero-padded batch (rows past each utterance's
    extent re-zeroed after every layer, temporal mean over its own rows) == each utterance's own forward, bitwise."""
    eng = Engine(emul_lib())
    n_stages = 2
    sd = O.make_state_dict(seed=29, num_classes=4, n_stages=n_stages)
    tsd = torch_sd(sd)
    pw = eng.pack_weights(tsd, n_stages, with_f16=True)
    folded = {n: eng.bn_fold(b) for n, b in make_bns(tsd, n_stages).items()}
    lens = [7, 16, 11, 1]
    T = 16
    rs = np.random.RandomState(4)
    x = torch.zeros(len(lens), 1, T, 64)
    for i, t in enumerate(lens):
        x[i, 0, :t] = torch.from_numpy(rs.randn(t, 64).astype(np.float32))
    e = eng.forward_eval_planned(x, pw, folded, precision=precision, lengths=torch.tensor(lens))
    e2 = eng.forward_eval_planned(x, pw, folded, precision=precision, lengths=torch.tensor(lens))      # plan re-use
    assert torch.equal(e, e2)
    for i, t in enumerate(lens):
        alone = eng.forward_eval(x[i:i + 1, :, :t].contiguous(), pw, folded, precision=precision)
        assert torch.equal(e[i:i + 1], alone), (precision, t)
    # without the masks the padding leaks: the unmasked padded forward differs for the short utterances
    plain = eng.forward_eval_planned(x, pw, folded, precision=precision)
    assert not torch.equal(plain[0], e[0]) and torch.equal(plain[1], e[1])
    with pytest.raises(ValueError):
        eng.forward_eval_planned(x, pw, folded, precision=precision, lengths=torch.tensor([7, 17, 11, 1]))


def test_enrolment_scores_over_sets_of_different_sizes():
    from deepspeaker_pytorch_amd import scoring
    scoring._engine_override = Engine(emul_lib())
    try:
        rs = np.random.RandomState(8)
        sizes = [3, 1, 5, 2]
        test = rs.randn(len(sizes), 64).astype(np.float32)
        enrol = rs.randn(sum(sizes), 64).astype(np.float32)
        got = scoring.enrolment_scores(torch.from_numpy(test), torch.from_numpy(enrol), sizes).numpy()
        off = np.concatenate([[0], np.cumsum(sizes)])
        for i in range(len(sizes)):
            d = O.pairwise_distance(np.repeat(test[i:i + 1], sizes[i], 0), enrol[off[i]:off[i + 1]])
            assert abs(got[i] - d.mean()) < 1e-5
        with pytest.raises(ValueError):
            scoring.enrolment_scores(torch.from_numpy(test), torch.from_numpy(enrol), [3, 1, 5, 3])
    finally:
        scoring._engine_override = None


@pytest.mark.parametrize("B", [1, 3, 4])
def test_small_batch_tail_equals_the_three_launch_tail(B):
    """ds_tail_small_f32 (serving latency: temporal mean + projection in one launch, then the norm) against the oracle and
    against ds_avgpool_time_f32 + ds_fc_l2norm_fwd_f32 on the same activations -- another summation order, f32 rounding."""
    eng = Engine(emul_lib())
    rs = np.random.RandomState(40 + B)
    hr, wc, c, n_out = 10, 4, 512, 512
    a = torch.from_numpy(np.abs(rs.randn(B, hr, wc, c)).astype(np.float32))
    wfc = torch.from_numpy((rs.randn(n_out, c * wc) / np.sqrt(c * wc)).astype(np.float32))     # reference order c*F + f
    bias = torch.from_numpy((rs.randn(n_out) * 0.1).astype(np.float32))
    k = wc * c
    rows = torch.empty(n_out * k)
    eng.lib.call("ds_pack_fc_weight_rows_f32", eng._p(wfc), eng._p(rows), n_out, c, wc, None)
    f, e = torch.empty(B, n_out), torch.empty(B, n_out)
    eng.lib.call("ds_tail_small_f32", eng._p(a), eng._p(rows), eng._p(bias), eng._p(f), eng._p(e), B, hr, k, n_out, 10.0, 1e-10, None)
    # oracle: NCHW mean over time, flatten c*F + f (model.py:207-213)
    x = a.numpy().transpose(0, 3, 1, 2)                     # [B, C, Hr, Wc]
    pooled = x.mean(axis=2).reshape(B, -1)
    f_ref = pooled @ wfc.numpy().T + bias.numpy()
    e_ref = O.l2_norm_scale(f_ref.astype(np.float64))
    assert rel_err(f.numpy(), f_ref) < 2e-6 and rel_err(e.numpy(), e_ref) < 2e-6
    # the three-launch tail
    packed = torch.empty(n_out * k)
    eng.lib.call("ds_pack_fc_weight_f32", eng._p(wfc), eng._p(packed), n_out, c, wc, None)
    pooled_t = torch.empty(B, k)
    eng.lib.call("ds_avgpool_time_f32", eng._p(a), eng._p(pooled_t), B, hr, wc, c, None)
    ws = torch.empty(eng.lib.raw("ds_fc_workspace_floats")(B, k, n_out))
    f2, e2 = torch.empty(B, n_out), torch.empty(B, n_out)
    eng.lib.call("ds_fc_l2norm_fwd_f32", eng._p(pooled_t), eng._p(packed), eng._p(bias), eng._p(ws), eng._p(f2), eng._p(e2), B, k,
                 n_out, 10.0, 1e-10, None)
    assert rel_err(e.numpy(), e2.numpy()) < 2e-6
    assert eng.lib.raw("ds_tail_small_f32")(eng._p(a), eng._p(rows), eng._p(bias), eng._p(f), eng._p(e), 5, hr, k, n_out, 10.0,
                                            1e-10, None) != 0           # B above DS_TAIL_SMALL_MAX_B is refused
