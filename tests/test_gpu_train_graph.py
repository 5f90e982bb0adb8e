"""The whole triplet training step as ONE HIP graph (deepspeaker-pytorch_amd/train_graph.py) on a real MI355X: replays must
walk the eager loop's trajectory, constructing the object must leave model and optimizer where they were, and the model
must come back usable (reference loop: train_triplet.py:215-224)."""
import numpy as np
import pytest
import torch

import deepspeaker_oracle as O

pytestmark = pytest.mark.gpu


def build(sd, tp, optimizer):
    from deepspeaker_pytorch_amd.model import DeepSpeakerModel
    from deepspeaker_pytorch_amd.optim import create_optimizer
    kw = dict(precision="f16", train_precision="f16") if tp == "f16" else dict(precision=tp)
    m = DeepSpeakerModel(512, 16, **kw)
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    m = m.cuda().train()
    return m, create_optimizer(m, 0.01, optimizer, lr_decay=1e-2)


@pytest.mark.parametrize("tp,optimizer", [("f16", "adagrad"), ("bf16x3", "sgd"), ("f16", "adam")])
def test_graphed_step_walks_the_eager_trajectory(tp, optimizer):
    from deepspeaker_pytorch_amd.model import TripletMarginLoss
    from deepspeaker_pytorch_amd.train_graph import GraphedTripletStep
    sd = O.make_state_dict(seed=41, num_classes=16)
    batches = [[torch.from_numpy(O.make_input(seed=700 + 10 * b + i, batch=8, frames=160)).cuda() for i in range(3)] for b in range(2)]
    steps = 6
    # eager
    m, opt = build(sd, tp, optimizer)
    loss_fn = TripletMarginLoss(0.1)
    eager = []
    for it in range(steps):
        out = m.forward_triplet(*batches[it % 2])
        loss = loss_fn.forward(*out)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        eager.append(float(loss))
    p_eager = {n: p.detach().clone() for n, p in m.named_parameters()}
    b_eager = {n: b.detach().clone() for n, b in m.named_buffers()}
    # graphed, from the same start
    m2, opt2 = build(sd, tp, optimizer)
    before = {n: p.detach().clone() for n, p in m2.named_parameters()}
    before_b = {n: b.detach().clone() for n, b in m2.named_buffers()}
    with GraphedTripletStep(m2, opt2, margin=0.1, example=batches[0]) as gstep:
        # constructing it (three real warm-up steps + the capture) left everything where it was
        assert all(torch.equal(p.detach(), before[n]) for n, p in m2.named_parameters())
        assert all(torch.equal(b.detach(), before_b[n]) for n, b in m2.named_buffers())
        assert all(float(v.abs().sum()) == 0.0 for st in opt2.state.values() for k, v in st.items()
                   if torch.is_tensor(v) and v.is_cuda)
        graphed = [float(gstep(*batches[it % 2])) for it in range(steps)]
    print(f"\n[{tp} {optimizer}] eager  :", " ".join(f"{v:.6f}" for v in eager))
    print(f"[{tp} {optimizer}] graphed:", " ".join(f"{v:.6f}" for v in graphed))
    np.testing.assert_allclose(graphed, eager, rtol=2e-5, atol=1e-7)
    worst = max(float((p.detach() - p_eager[n]).abs().max() / p_eager[n].abs().max().clamp_min(1e-12))
                for n, p in m2.named_parameters() if n in p_eager and not n.startswith("model.classifier"))
    assert worst < 1e-5, worst
    for n, b in m2.named_buffers():                     # running statistics and num_batches_tracked advanced on the device
        assert torch.allclose(b.float(), b_eager[n].float(), rtol=1e-5, atol=1e-6), n
    # after close(): host bookkeeping is up to date, the model evaluates with the trained weights
    if optimizer != "sgd":
        assert all(float(st["step"]) == steps for st in opt2.state.values() if "step" in st)
    x = batches[0][0]
    with torch.no_grad():
        e1, e2 = m.eval()(x).clone(), m2.eval()(x).clone()
    assert float((e1 - e2).abs().max()) < 1e-3 * float(e1.abs().max())
