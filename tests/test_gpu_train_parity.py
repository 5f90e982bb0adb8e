"""The TRAINING step (train_triplet.py:215-223: three train-mode forwards, TripletMarginLoss, backward) on a real
MI355X, pinned two ways:

* against the unmodified reference at the bench configuration itself -- 3 x 256 utterances of BASELINE configs[1],
  tests/golden/reference_cfg1_train.npz (float32 run and float64 run of /root/reference/model.py, made by
  tests/golden/make_golden.py): loss, all 768 train-mode embeddings, the 36 running statistics, every gradient;
* against the torch restatement of that step (oracle/torch_restatement.py, itself pinned bit-for-bit to the reference's
  forward and to 2e-5 on its gradients by tests/test_oracle_golden.py) evaluated WITH THE HIP FORWARD'S OWN clipped-ReLU
  masks: both sides then differentiate the same piecewise-linear function and every one of the 38 gradient tensors must
  agree to 1e-4 -- at B = 8, at 3 x 64 and at the 768-row bench size (measured: 9e-6 exact-f32, 7e-5 split-bf16) --
  instead of the 1e-2 .. 8e-2 that masks flipped by rounding cost an unmasked comparison (the reference's own fp32 and
  fp64 runs differ by 4e-3 at this size for that reason alone).
"""
import os

import numpy as np
import pytest
import torch

import deepspeaker_oracle as O
import torch_restatement as TR
from conftest import ROOT, rel_err

pytestmark = pytest.mark.gpu


def build(sd, precision, num_classes):
    from deepspeaker_pytorch_amd.model import DeepSpeakerModel
    m = DeepSpeakerModel(512, num_classes, precision=precision)
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    return m.cuda().train()


def hip_step(m, xs, margin=0.1):
    """forward_triplet + loss + backward; returns (loss, embeddings, masks per member as NCHW bool, grads, model)"""
    from deepspeaker_pytorch_amd.model import TripletMarginLoss
    outs = m.forward_triplet(*xs)
    saved = outs[0].grad_fn.saved_forward                  # the autograd node's saved activations (channels-last)
    bm = xs[0].shape[0]
    masks = []
    for g in range(3):
        d = {}
        for key, act in saved.acts.items():
            a = act[g * bm:(g + 1) * bm]
            d[key] = ((a > 0) & (a < 20)).permute(0, 3, 1, 2).contiguous().cpu()      # the backward kernels' rule
        masks.append(d)
    loss = TripletMarginLoss(margin).forward(*outs)
    m.zero_grad()
    loss.backward()
    grads = {n: p.grad.detach().cpu() for n, p in m.named_parameters() if p.grad is not None}
    return float(loss.detach()), [o.detach().cpu() for o in outs], masks, grads


def rel_l2(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


# (members' batch, frames, oracle dtype, bar): float64 where the host forward+backward stays small; the 768-row bench
# size runs the oracle in float32 (oneDNN, ~20 s on the GPU box's host; 6 GB) -- its own rounding is ~1e-6
CASES = [(8, 160, torch.float64), (64, 160, torch.float64), (256, 160, torch.float32)]


@pytest.mark.parametrize("precision", ["f32", "bf16x3"])
@pytest.mark.parametrize("bm,frames,odt", CASES)
def test_training_step_gradients_vs_masked_oracle(precision, bm, frames, odt):
    torch.set_num_threads(min(64, os.cpu_count() or 8))
    ncls = 1211 if bm == 256 else 16
    sd = O.make_state_dict(seed=0 if bm == 256 else 31, num_classes=ncls)
    if bm == 256:       # BASELINE configs[1]: the very batch bench.py's train_step times
        x = torch.randn(768, 1, 160, 64, generator=torch.Generator(device="cpu").manual_seed(1234))
        xs_cpu = [x[i * 256:(i + 1) * 256].contiguous() for i in range(3)]
    else:
        xs_cpu = [torch.from_numpy(O.make_input(seed=32 + i, batch=bm, frames=frames)) for i in range(3)]
    m = build(sd, precision, ncls)
    loss, embs, masks, grads = hip_step(m, [x.cuda() for x in xs_cpu])
    tsd = {k: torch.from_numpy(np.array(v)) for k, v in sd.items()}
    ref = TR.triplet_train_step(tsd, xs_cpu, 0.1, masks=masks, dtype=odt)
    assert abs(loss - float(ref["loss"])) <= (2e-5 if precision == "f32" else 1e-4) * abs(float(ref["loss"]))
    for e, r in zip(embs, ref["embeddings"]):
        assert rel_err(e.numpy(), r.float().numpy()) < 3e-5
    # the oracle's own masks differ from the HIP forward's in a handful of boundary elements at most
    flips = sum(int((masks[g][k] != ((a > 0) & (a < 20))).sum()) for g in range(3) for k, a in ref["acts"][g].items())
    total = sum(v.numel() for d in masks for v in d.values())
    worst = {}
    for name, g in grads.items():
        assert name in ref["grads"], name
        worst[name] = rel_l2(g, ref["grads"][name])
    assert len(worst) == 38
    top = sorted(worst.items(), key=lambda kv: -kv[1])[:3]
    print(f"\n[{precision} {3 * bm} rows] loss {loss:.7f}; clip masks differing from the oracle's own forward: {flips} of "
          f"{total}; worst gradient rel-L2: " + ", ".join(f"{k} {v:.1e}" for k, v in top))
    assert max(worst.values()) < 1e-4, top
    # ... and that handful is BOUNDED, not just printed: measured at 768 rows 222 (exact f32) / 2773 (bf16x3) of 7e8
    # elements, i.e. 3e-7 / 4e-6 of them; a kernel regression that flips several times more must not hide inside the
    # masked comparison above (which would simply adopt the flipped masks)
    flip_cap = max(16, int(total * (1.5e-6 if precision == "f32" else 2e-5)))
    assert flips <= flip_cap, (flips, flip_cap, total)
    # running statistics: three updates in call order
    for k, v in ref["running"].items():
        got = dict(m.state_dict())[k].cpu()
        assert rel_err(got.numpy(), v.float().numpy()) < 2e-5, k
    assert int(m.model.bn1.num_batches_tracked) == 3


@pytest.fixture(scope="module")
def cfg1t():
    g = np.load(os.path.join(ROOT, "tests", "golden", "reference_cfg1_train.npz"))
    x = torch.randn(768, 1, 160, 64, generator=torch.Generator(device="cpu").manual_seed(1234))
    return g, O.make_state_dict(seed=0, num_classes=1211), x


def grad_digest(a):
    a = np.asarray(a, np.float64).ravel()
    stride = max(1, a.size // 64)
    return np.concatenate([[np.sqrt((a * a).sum()), a.sum()], a[:16], a[::stride][:64]])


@pytest.mark.parametrize("precision", ["f32", "bf16x3"])
def test_training_step_bench_size_vs_reference_golden(cfg1t, precision):
    """The 768-row step against the UNMODIFIED reference's recorded step (float32 and float64 runs)."""
    g, sd, x = cfg1t
    m = build(sd, precision, 1211)
    loss, embs, _, grads = hip_step(m, [x[i * 256:(i + 1) * 256].contiguous().cuda() for i in range(3)])
    e = torch.cat(embs).numpy()
    for tag in ("", "64"):
        ref_loss = float(g[f"cfg1t{tag}_loss"])
        assert abs(loss - ref_loss) <= 2e-5 * abs(ref_loss), (tag, loss, ref_loss)
        assert rel_err(e, g[f"cfg1t{tag}_emb"]) < 3e-5
    sdm = dict(m.state_dict())
    for k in g.files:
        if k.startswith("cfg1t64_stat/"):
            name = k.split("/", 1)[1]
            if "num_batches" in name:
                assert int(sdm[name]) == int(g[k]) == 3
            else:
                assert rel_err(sdm[name].cpu().numpy(), g[k]) < 2e-5, name
    # gradients: digests of all 38 tensors, and the small tensors in full, against the float64 reference run (the exact
    # derivative); clip masks that round differently between this forward and the reference's float64 forward are what
    # is left (each moves one utterance's contribution: 1/768 of a tensor's gradient mass)
    worst_d, worst_f = {}, {}
    for name, gr in grads.items():
        ref = g["cfg1t64_grad/" + name]
        worst_d[name] = float(np.abs(grad_digest(gr.numpy()) - ref).max() / np.abs(ref).max())
        fk = "cfg1t64_gfull/" + name
        if fk in g.files:
            worst_f[name] = rel_l2(gr, torch.from_numpy(g[fk]))
    assert len(worst_d) == 38 and len(worst_f) >= 25
    # the reference's own float32 run against its float64 run, same measure: the yardstick
    ref32 = max(float(np.abs(g["cfg1t_grad/" + n] - g["cfg1t64_grad/" + n]).max() / np.abs(g["cfg1t64_grad/" + n]).max())
                for n in grads)
    print(f"\n[{precision}] loss {loss:.7f} (reference {float(g['cfg1t_loss']):.7f}); gradient digests vs the float64 "
          f"reference: worst {max(worst_d.values()):.2e} (the reference's own float32 run: {ref32:.2e}); small tensors in "
          f"full, worst rel-L2 {max(worst_f.values()):.2e}")
    # Unmasked, a gradient comparison at this size measures how many clip masks round differently, not the kernels: the
    # reference's own float32 run is 4e-3 from its float64 run on this measure, the exact-f32 path 1.2e-2 (222 of 7e8
    # masks differ), the split-bf16 path 5e-2 (2773).  The tight gradient bars at this size are the masked-oracle test's
    # (1e-5 / 7e-5 above); here the band is asserted.
    bar_d, bar_f = (5 * max(ref32, 1e-3), 1e-2) if precision == "f32" else (8e-2, 2e-2)
    assert max(worst_d.values()) < bar_d and max(worst_f.values()) < bar_f


def test_member_streams_forward_equals_three_calls_bitwise():
    """Engine._forward_train_group_streams (one HIP stream per member of the triplet step) launches, per member, exactly
    the kernels of a separate `model(x)` call on that member's slice: embeddings and running statistics (three momentum
    updates in call order) must be BITWISE those of the reference's call pattern `model(a), model(p), model(n)` -- which
    also makes this the race detector of the stream choreography -- and equal the lock-step forward over one batch up
    to the order in which per-tile statistics are folded; gradients of the one grouped backward pass against three
    accumulated backward passes to summation-order rounding."""
    from deepspeaker_pytorch_amd.engine import Engine
    from deepspeaker_pytorch_amd.model import TripletMarginLoss
    sd = O.make_state_dict(seed=31, num_classes=16)
    xs = [torch.from_numpy(O.make_input(seed=32 + i, batch=64, frames=160)).cuda() for i in range(3)]
    res = {}
    try:
        for mode in ("three_calls", "lock_step", "streams", "streams_again"):
            Engine.MEMBER_STREAMS = mode.startswith("streams")
            m = build(sd, "bf16x3", 16)
            outs = (m(xs[0]), m(xs[1]), m(xs[2])) if mode == "three_calls" else m.forward_triplet(*xs)
            loss = TripletMarginLoss(0.1).forward(*outs)
            loss.backward()
            torch.cuda.synchronize()
            res[mode] = ([o.detach().clone() for o in outs], {k: v.clone() for k, v in m.state_dict().items()},
                         {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None})
    finally:
        Engine.MEMBER_STREAMS = True
    for other in ("three_calls", "streams_again"):
        for x_, y_ in zip(res["streams"][0], res[other][0]):
            assert torch.equal(x_, y_), other
        for k, v in res["streams"][1].items():
            assert torch.equal(v, res[other][1][k]), (other, k)
    for k, v in res["streams"][2].items():
        assert torch.equal(v, res["streams_again"][2][k]), k
        # (not against the lock-step pass: its statistics differ in the last bit, which flips a clip mask or two --
        # 3e-3 on conv1's gradient at this size; the masked-oracle test above is where gradients are held tight)
        assert rel_l2(v, res["three_calls"][2][k]) < 2e-5, (k, rel_l2(v, res["three_calls"][2][k]))
    # lock-step: statistics folded from differently shaped tiles -> last-bit differences in scale / shift, which the
    # hi / lo operand split of the next layer turns into ~2^-16 per product (measured 4.6e-6 on the embeddings)
    for x_, y_ in zip(res["streams"][0], res["lock_step"][0]):
        assert rel_err(x_.cpu().numpy(), y_.cpu().numpy()) < 2e-5
    for k, v in res["streams"][1].items():
        if v.is_floating_point():
            assert rel_err(v.cpu().numpy(), res["lock_step"][1][k].cpu().numpy()) < 2e-5, k


@pytest.mark.parametrize("bm,frames", [(64, 160), (5, 37)])
def test_fused_batchnorm_backward_reductions_equal_the_two_step_sequence(bm, frames):
    """backward._dgrad_bn_bwd / _dgrad_s2_bn_bwd (the BatchNorm-backward reduction and the clipped-ReLU mask inside the
    3x3 / 5x5-stride-2 data-gradient kernels' epilogues) against the separate data-gradient + reduction launches they
    replace, on the GPU, for a batch of three members: the fused entry points are taken, and every gradient agrees to
    the rounding of a differently ordered sum (the mask itself is bit-identical: same fma, same inputs).  (5, 37): odd
    sizes -- ragged tiles, parity classes of different extents; there the planner decides which layers fuse."""
    from deepspeaker_pytorch_amd import backward
    from deepspeaker_pytorch_amd.model import TripletMarginLoss, get_engine
    sd = O.make_state_dict(seed=31, num_classes=16)
    xs = [torch.from_numpy(O.make_input(seed=32 + i, batch=bm, frames=frames)).cuda() for i in range(3)]
    lib = get_engine().lib
    out = {}
    try:
        for fuse in (True, False):
            backward.FUSE_DGRAD_BN_BWD = fuse
            m = build(sd, "bf16x3", 16)
            lib.trace = {}
            loss = TripletMarginLoss(0.1).forward(*m.forward_triplet(*xs))
            loss.backward()
            torch.cuda.synchronize()
            n3, n5 = lib.trace.get("ds_conv_dgrad_bnbwd_bf16", 0), lib.trace.get("ds_conv_dgrad_s2_bnbwd_bf16", 0)
            lib.trace = None
            if fuse:
                assert n3 >= 4 and n5 >= 1, (n3, n5)
                print(f"\n[{3 * bm} x {frames} frames] fused reductions: {n3} of 8 3x3 layers, {n5} of 3 stride-2 layers")
            else:
                assert n3 == 0 and n5 == 0
            out[fuse] = {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}
    finally:
        backward.FUSE_DGRAD_BN_BWD = True
        lib.trace = None
    assert set(out[True]) == set(out[False]) and len(out[True]) == 38
    for k, v in out[False].items():
        assert rel_l2(out[True][k], v) < 1e-5, (k, rel_l2(out[True][k], v))
