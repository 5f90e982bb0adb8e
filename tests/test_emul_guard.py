"""The fp16 precision guard (deepspeaker-pytorch_amd/precision_guard.py) on the host emulator: the decision logic and the
model-level wiring, with the unmodified kernels computing the fp16 and f32-class sample embeddings it compares.  CPU
tensors reach the emulated library here ONLY to test host logic without a GPU (the product refuses them)."""
import numpy as np
import pytest
import torch

import deepspeaker_oracle as O
from conftest import rel_err
from emul_util import emul_lib


@pytest.fixture()
def emul_model(monkeypatch):
    from deepspeaker_pytorch_amd import model as M
    from deepspeaker_pytorch_amd.engine import Engine
    monkeypatch.setattr(M, "_engine", Engine(emul_lib()))
    monkeypatch.setattr(M, "_require_cuda", lambda t, what: None)

    def build(threshold, n_stages=2, seed=3):
        sd = O.make_state_dict(seed=seed, num_classes=4, n_stages=n_stages)
        m = M.DeepSpeakerModel(512, 4, n_stages=n_stages, precision="f16", f16_guard=threshold)
        m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
        return m.eval(), sd
    return build


def test_guard_measures_and_escalates(emul_model):
    x = torch.from_numpy(O.make_input(seed=8, batch=3, frames=24))
    # (a) a threshold the fp16 path cannot meet: the first forward measures, escalates, and returns f32-class embeddings
    m, sd = emul_model(1e-6)
    with torch.no_grad():
        e = m(x).clone()
    g = m.f16_guard
    assert g.verdict == "bf16x3" and g.checks == 1 and g.escalations == 1 and g.source == "check"
    assert 1e-6 < g.sample_error < 2e-3 and abs(g.estimate - 1.25 * g.sample_error) < 1e-12
    ref = O.forward(sd, x.numpy(), n_stages=2, dtype=np.float64)
    assert rel_err(e.numpy(), ref) < 5e-5
    # ... the same weights again: no second measurement
    with torch.no_grad():
        m(x)
    assert g.checks == 1 and g.calls == 2
    # (b) a threshold it meets: fp16 embeddings (bitwise those of an unguarded model), the verdict stays
    m2, _ = emul_model(0.5)
    m3, _ = emul_model(None)
    assert m3.f16_guard is None
    with torch.no_grad():
        e2, e3 = m2(x).clone(), m3(x).clone()
    assert m2.f16_guard.verdict == "f16" and m2.f16_guard.checks == 1 and torch.equal(e2, e3)
    assert 1e-6 < rel_err(e2.numpy(), ref) < 1e-3
    assert abs(m2.f16_guard.sample_error - g.sample_error) < 1e-9      # the same measurement in both models
    rep = m2.f16_guard.report()
    assert rep["verdict"] == "f16" and rep["threshold"] == 0.5 and rep["forwards"] == 1


def test_guard_rechecks_on_new_weights_rate_limited_and_takes_probe_samples(emul_model):
    x = torch.from_numpy(O.make_input(seed=9, batch=2, frames=16))
    m, _ = emul_model(0.5)
    g = m.f16_guard
    g.min_gap = 3
    with torch.no_grad():
        m(x)
        assert g.checks == 1
        m.model.conv1.weight.data.mul_(1.0)             # (no version bump through .data: nothing to see)
        m(x)
        assert g.checks == 1
        def bump():
            with torch.no_grad():
                m.model.conv1.weight.mul_(1.0)          # a new weight generation
        bump()
        m(x)
        assert g.checks == 2                            # ONE change after a stable generation: measured at once
        for _ in range(2):                              # the weights change on EVERY forward (evaluation interleaved with
            bump()                                      # optimizer steps): rate limited, the verdict stands
            m(x)
            assert g.checks == 2
        bump()
        m(x)
        assert g.checks == 3                            # 3 forwards after the last check: measured again
        bump()
        m(x)
        assert g.checks == 3
        m.load_state_dict(m.state_dict())               # a checkpoint load re-arms the check whatever the rate limit says
        m(x)
        assert g.checks == 4
        m.train()
        m.eval()                                        # ... and so does coming back from training (with new weights)
        bump()
        m(x)
        assert g.checks == 5
    # the refinement's probes feed the same decision, upwards only
    from deepspeaker_pytorch_amd.mining import refine_policy
    pol = refine_policy(m)
    assert pol.guard is g
    pol.observe(0, 1e-4, 4, emb_err=1e-4)
    assert g.verdict == "f16"
    pol.observe(0, 1e-4, 4, emb_err=0.45)               # x 1.3 > 0.5
    assert g.verdict == "bf16x3" and g.source == "probes" and g.escalations == 1
    pol.observe(0, 1e-4, 4, emb_err=0.0)                # escalated: the probes compare f32-class with itself
    assert g.verdict == "bf16x3"
    with torch.no_grad():
        for _ in range(3):
            m(x)                                        # new check (same weights, but a synchronous one is due only on new
        assert g.verdict == "bf16x3"                    # weights): stays escalated until a check says otherwise
        with torch.no_grad():
            m.model.conv1.weight.mul_(1.0)
        m(x)
    assert g.verdict == "f16" and g.deescalations == 1
