"""Concurrency of the persistent fp16 kernels on a real MI355X: their device-side tile counters ("scheduling slots",
csrc/ds_device.h) must be private to whatever is in flight together -- two streams racing each other with more launches
in between than a stream's slot ring holds, and a captured graph replaying next to eager launches."""
import numpy as np
import pytest
import torch

import deepspeaker_oracle as O
from conftest import perf_note

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


def test_cold_capture_first_persistent_launch_inside_a_graph():
    """the library owns no device memory and never synchronises, so a stream capture whose FIRST persistent
    launch was never warmed up must work: the scheduler workspace is the wrapper's (allocated through torch when the model
    is moved to the device).  Fresh process (the slot table is per process): capture the very first fp16 forward, replay it,
    compare with the eager forward; then an eager forward on a stream the process has not used yet."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys, numpy as np, torch
sys.path[:0] = [%r, %r]
import deepspeaker_oracle as O
from deepspeaker_pytorch_amd.model import DeepSpeakerModel, get_engine
sd = O.make_state_dict(seed=11, num_classes=16)
m = DeepSpeakerModel(512, 16, precision="f16", f16_guard=None)
m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
m = m.cuda().eval()
lib = get_engine().lib
free0 = int(lib.raw("ds_sched_free_slots")())
assert free0 >= 64, free0                                   # handed over by .cuda(), before any launch
x = torch.from_numpy(O.make_input(seed=905, batch=32, frames=160)).cuda()
sx = x.clone()
g = torch.cuda.CUDAGraph()
with torch.no_grad(), torch.cuda.graph(g):                  # no warm-up of any kind
    se = m(sx)
free1 = int(lib.raw("ds_sched_free_slots")())
assert free0 - 9 <= free1 < free0, (free0, free1)           # every captured persistent launch keeps one slot for good
g.replay()
torch.cuda.synchronize()
got = se.clone()
with torch.no_grad():
    ref = m(x).clone()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        ref2 = m(x).clone()
torch.cuda.synchronize()
assert torch.equal(got, ref) and torch.equal(ref2, ref)
g.replay(); torch.cuda.synchronize()
assert torch.equal(se, ref)
print("cold capture ok", free0, int(lib.raw("ds_sched_free_slots")()))
""" % (root, os.path.join(root, "oracle"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "cold capture ok" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def test_launch_bound_events_read_the_kernel_and_leave_results_alone():
    """Per-launch timing of the bench's live roofline: events bound to the launch itself (ds_launch_timing_arm ->
    hipExtLaunchKernelGGL) instead of an event pair recorded around it.  Same results bit for bit, one timed launch per
    call, durations that agree with the event-pair measurement of the same launches (which contains them)."""
    from deepspeaker_pytorch_amd.model import get_engine
    sd = O.make_state_dict(seed=11, num_classes=16)
    m = build(sd)
    x = torch.from_numpy(O.make_input(seed=903, batch=192, frames=160)).cuda()
    eng = get_engine()
    with torch.no_grad():
        ref = m(x).clone()
        torch.cuda.synchronize()
        out = {}
        for bound in (True, False):
            eng.self_timed_launches = bound
            for _ in range(3):
                m(x)
            torch.cuda.synchronize()
            eng.profile = []
            e = m(x).clone()
            torch.cuda.synchronize()
            prof, eng.profile = eng.profile, None
            assert torch.equal(e, ref)
            out[bound] = [(label, e0.elapsed_time(e1)) for label, _, e0, e1, prec in prof if prec == "f16"]
        eng.self_timed_launches = True
    assert [l for l, _ in out[True]] == [l for l, _ in out[False]] and len(out[True]) == 9
    from deepspeaker_pytorch_amd.engine import LaunchEvent
    for (label, t_bound), (_, t_pair) in zip(out[True], out[False]):
        print(f"{label:32s} launch-bound {t_bound * 1e3:7.1f} us   event pair {t_pair * 1e3:7.1f} us")
        assert t_bound > 0 and t_pair > 0
        perf_note(0.003 < t_bound < 5.0 and t_bound < t_pair * 1.25 + 0.01, (label, t_bound, t_pair))
    perf_note(sum(t for _, t in out[True]) <= sum(t for _, t in out[False]) * 1.1, "launch-bound sum vs event-pair sum")
    # an armed pair that no launch consumed is simply dropped
    a, b = LaunchEvent(eng.lib), LaunchEvent(eng.lib)
    eng.lib.call("ds_launch_timing_arm", a.handle, b.handle)
    assert eng.lib.raw("ds_launch_timing_end")() == 0
    import time
    t0 = time.perf_counter()
    with pytest.raises(Exception):              # ... and reading it is an error, not a hang (bench.py falls back to pairs)
        a.elapsed_time(b)
    perf_note(time.perf_counter() - t0 < 5.0, "reading an unconsumed launch event returns at once")


def test_batches_in_flight_yield_the_sequential_embeddings_in_order():
    """pipeline.BatchesInFlight: consecutive batches alternating over two streams -- the tensors model(x) returns, bit for
    bit and in order; different batch sizes in one sequence; a consumer that reads each result at once."""
    from deepspeaker_pytorch_amd.pipeline import BatchesInFlight
    sd = O.make_state_dict(seed=11, num_classes=16)
    m = build(sd)
    sizes = [96, 96, 48, 96, 192, 48, 96]
    xs = [torch.from_numpy(O.make_input(seed=950 + i, batch=b, frames=160)).cuda() for i, b in enumerate(sizes)]
    with torch.no_grad():
        want = [m(x).clone() for x in xs]
    torch.cuda.synchronize()
    for n in (1, 2, 3):
        got = [e.clone() for e in BatchesInFlight(m, in_flight=n)(xs)]
        torch.cuda.synchronize()
        assert len(got) == len(want)
        for a, b in zip(got, want):
            assert torch.equal(a, b)
    pipe = BatchesInFlight(m)
    sums = [float(e.sum()) for e in pipe(xs)]                  # the consumer synchronises on every result
    assert sums == [float(w.sum()) for w in want]
    m.train()
    with pytest.raises(RuntimeError):
        list(pipe(xs[:1]))
