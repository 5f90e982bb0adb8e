"""Concurrency of the persistent fp16 kernels on a real MI355X: their device-side tile counters ("scheduling slots",
csrc/ds_device.h) must be private to whatever is in flight together -- two streams racing each other with more launches
in between than a stream's slot ring holds, and a captured graph replaying next to eager launches."""
import numpy as np
import pytest
import torch

import deepspeaker_oracle as O

pytestmark = pytest.mark.gpu


def build(sd, precision="f16"):
    from deepspeaker_pytorch_amd.model import DeepSpeakerModel
    m = DeepSpeakerModel(512, 16, precision=precision)
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    return m.cuda().eval()


def test_f16_forwards_racing_on_two_streams_are_bitwise_the_serial_ones():
    sd = O.make_state_dict(seed=11, num_classes=16)
    m = build(sd)
    xa = torch.from_numpy(O.make_input(seed=901, batch=192, frames=160)).cuda()
    xb = torch.from_numpy(O.make_input(seed=902, batch=48, frames=160)).cuda()
    with torch.no_grad():
        ref_a, ref_b = m(xa).clone(), m(xb).clone()
    torch.cuda.synchronize()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    outs_a, outs_b = [], []
    with torch.no_grad():
        for st_, x in ((sa, xa), (sb, xb)):         # each stream's launch plan (and its activation buffers) up front
            with torch.cuda.stream(st_):
                m(x)
        torch.cuda.synchronize()
        for it in range(24):
            # one big forward queued on A, then a burst of small ones on B: far more persistent launches (9 per
            # forward) than a ring of slots holds are enqueued on B while A's are still queued or running
            with torch.cuda.stream(sa):
                outs_a.append(m(xa).clone())
            with torch.cuda.stream(sb):
                for _ in range(4):
                    outs_b.append(m(xb).clone())
    torch.cuda.synchronize()
    for e in outs_a:
        assert torch.equal(e, ref_a)
    for e in outs_b:
        assert torch.equal(e, ref_b)


def test_graph_replay_next_to_eager_f16_launches():
    sd = O.make_state_dict(seed=11, num_classes=16)
    m = build(sd)
    xa = torch.from_numpy(O.make_input(seed=903, batch=64, frames=160)).cuda()
    xb = torch.from_numpy(O.make_input(seed=904, batch=96, frames=160)).cuda()
    with torch.no_grad():
        ref_a, ref_b = m(xa).clone(), m(xb).clone()
    g = m.graphed(xa)
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    got_a, got_b = [], []
    with torch.no_grad():
        with torch.cuda.stream(side):
            m(xb)
        torch.cuda.synchronize()
        for it in range(16):
            got_a.append(g(xa).clone())                 # replay on the current stream ...
            with torch.cuda.stream(side):               # ... while eager launches of the same kernels run beside it
                for _ in range(3):
                    got_b.append(m(xb).clone())
    torch.cuda.synchronize()
    for e in got_a:
        assert torch.equal(e, ref_a)
    for e in got_b:
        assert torch.equal(e, ref_b)
