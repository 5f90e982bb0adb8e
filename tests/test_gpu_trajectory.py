"""BASELINE configs[3] in miniature (VERDICT r3 "next" #6): "training to convergence" is more than one step deep.
The triplet-regime loop of train_triplet.py:215-224 -- train-mode forwards of a / p / n, TripletMarginLoss, backward,
plain SGD -- for 30 steps on 10 synthetic speakers, then the test-time scoring of train_triplet.py:337-366 with the
threshold sweep of eval_metrics.calculate_roc and the EER, on a real MI355X against the same loop run through the
UNMODIFIED reference on the CPU (tests/golden/reference_trajectory.npz, made by make_golden.py).

The loop is chaotic -- a hinge or a clip mask that rounds the other way changes which triplets / elements carry gradient,
and SGD amplifies it -- so the golden also records how far the reference is from ITSELF when only its summation order
changes (3 threads instead of 8: 1.6 % at step 5, 28 % at step 11, 100 % at step 21) or it runs in float64 (EER 0.223 /
0.183 / 0.250 for the three runs).  The bars: the first steps tight (where nothing has been amplified yet), later steps
in windows whose median divergence stays within 3x the reference's own, the curve's level and trend, the EER inside the reference's own
spread +- 5 points."""
import os
import sys

import numpy as np
import pytest
import torch

import deepspeaker_oracle as O
from conftest import ROOT

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


@pytest.mark.parametrize("precision,loss_tol", [("f32", 0.02), ("bf16x3", 0.02), ("f16t", 0.05)])
def test_thirty_step_trajectory_and_eer_vs_reference(precision, loss_tol):
    from deepspeaker_pytorch_amd import scoring
    from deepspeaker_pytorch_amd.model import DeepSpeakerModel, PairwiseDistance, TripletMarginLoss
    from deepspeaker_pytorch_amd.optim import FusedSGD
    g = np.load(os.path.join(ROOT, "tests", "golden", "reference_trajectory.npz"))
    # (the batch builders live next to the golden maker; they use the oracle's seeded generators only)
    TRAJ = dict(param_seed=7, corpus_seed=21, speakers=10, utts=24, train_utts=20, frames=160, mix=(0.3, 0.2, 0.9),
                triplets=16, steps=30, lr=0.003, margin=0.1, triplet_seed0=1000)
    corpus = O.make_speaker_corpus(TRAJ["corpus_seed"], TRAJ["speakers"], TRAJ["utts"], TRAJ["frames"], mix=TRAJ["mix"])
    sd = O.make_state_dict(seed=TRAJ["param_seed"], num_classes=TRAJ["speakers"], randomize_bn=False)
    # "f16t": the opt-in fp16 training step (train_f16.py) under the same loop
    m = (DeepSpeakerModel(512, TRAJ["speakers"], precision="f16", train_precision="f16") if precision == "f16t"
         else DeepSpeakerModel(512, TRAJ["speakers"], precision=precision))
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    m = m.cuda().train()
    opt = FusedSGD(m.parameters(), lr=TRAJ["lr"], momentum=0.0, dampening=0.0, weight_decay=0.0)
    losses = []
    for it in range(TRAJ["steps"]):
        a, p, n, _, _ = O.sample_triplets(TRAJ["triplet_seed0"] + it, TRAJ["speakers"], TRAJ["train_utts"], TRAJ["triplets"])
        xs = [torch.from_numpy(O.gather_utterances(corpus, i)).cuda() for i in (a, p, n)]
        out_a, out_p, out_n = m(xs[0]), m(xs[1]), m(xs[2])                   # three calls, as train_triplet.py:215
        loss = TripletMarginLoss(TRAJ["margin"]).forward(out_a, out_p, out_n)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss))
    losses = np.array(losses)
    ref = g["traj_loss"]
    floor = np.maximum(ref, 5e-3)                   # (a hinge mean near 0 is compared on an absolute floor)
    rel = np.abs(losses - ref) / floor
    own = np.maximum(np.abs(g["traj_loss_alt_threads"] - ref), np.abs(g["traj_loss_f64"] - ref)) / floor
    print(f"\n[{precision}] per-step loss: reference, HIP, rel. difference, the reference's own divergence (threads / float64):\n"
          + "\n".join(f"  {i:2d} {r:.6f} {v:.6f}  {e:.2e}  {o:.2e}" for i, (r, v, e, o) in enumerate(zip(ref, losses, rel, own))))
    assert rel[:4].max() < loss_tol, rel[:4]                   # nothing amplified yet: the arithmetic itself
    # afterwards the divergence is a random walk on both sides: windows of steps, median against the reference's own
    for lo, hi in ((4, 10), (10, 20), (20, 30)):
        assert np.median(rel[lo:hi]) <= 3.0 * np.median(own[lo:hi]) + loss_tol, (lo, hi, np.median(rel[lo:hi]), np.median(own[lo:hi]))
    assert rel.max() < 3.0
    # level and trend of the curve (noisy: a fresh batch every step)
    assert abs(losses[20:].mean() - ref[20:].mean()) < 0.5 * ref[20:].mean()
    assert losses[20:].mean() < 0.8 * losses[:10].mean()
    # test-time scoring on held-out utterances of the same speakers, eval mode
    idx = np.array([(s_, u) for s_ in range(TRAJ["speakers"]) for u in range(TRAJ["train_utts"], TRAJ["utts"])], np.int64)
    x_test = torch.from_numpy(O.gather_utterances(corpus, idx)).cuda()
    ii, jj = np.triu_indices(len(idx), 1)
    same = idx[ii, 0] == idx[jj, 0]
    np.testing.assert_array_equal(same, g["traj_test_same"])
    m.eval()
    with torch.no_grad():
        e = m(x_test).clone()
        d = PairwiseDistance(2).forward(e[torch.from_numpy(ii).cuda()].contiguous(), e[torch.from_numpy(jj).cuda()].contiguous())
    v = scoring.evaluate(d, torch.from_numpy(same.astype(np.int32)).cuda())
    emb_err = float(np.abs(e.cpu().numpy() - g["traj_test_emb"]).max() / np.abs(g["traj_test_emb"]).max())   # (reported only)
    print(f"[{precision}] after 30 steps: test embeddings max|d|/max {emb_err:.3e}; EER {v.eer:.4f} (reference "
          f"{float(g['traj_eer']):.4f}; before training {float(g['traj_eer_before']):.4f}); best-threshold tpr/fpr/acc "
          f"{v.tpr:.4f}/{v.fpr:.4f}/{v.accuracy:.4f} (reference {g['traj_roc_tpr_fpr_acc']})")
    eers = [float(g[k]) for k in ("traj_eer", "traj_eer_alt_threads", "traj_eer_f64")]
    assert min(eers) - 0.05 <= v.eer <= max(eers) + 0.05, (v.eer, eers)     # inside the reference's own spread +- 5 points
