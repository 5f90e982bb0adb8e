"""CPU-side checks of the boundary: the HIP shared library loads and exports every symbol that
include/deepspeaker_hip.h declares (no compute is launched without a GPU), the Python surface mirrors
the reference's, and the product path refuses to run without a GPU instead of falling back."""
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "deepspeaker_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ds_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    from deepspeaker_pytorch_amd import _native
    assert header_symbols() == _native.exported_symbols()


def test_hip_library_exports_every_declared_symbol():
    from deepspeaker_pytorch_amd import _native
    path = os.path.join(ROOT, "deepspeaker-pytorch_amd", _native.LIB_NAME)
    if not os.path.exists(path):
        import __graft_entry__
        __graft_entry__.build()
    lib = _native.NativeLib(path)            # resolves every symbol or raises AttributeError
    assert lib.raw("ds_version")() >= 600            # round 6 (include/deepspeaker_hip.h)
    # the library owns no device memory: the persistent kernels' scheduler workspace is the caller's
    assert lib.raw("ds_sched_workspace_bytes")() == 1024 * 64
    assert lib.raw("ds_sched_set_workspace")(None, 65536) == -3 and lib.error_string(-5).startswith("no free tile-scheduling slot")
    assert lib.error_string(-1) == "bad shape"
    # argument validation happens before any launch, so it is testable without a GPU
    shp = _native.ConvShape(1, 8, 8, 7, 64, 3, 1)
    assert lib.raw("ds_conv_stats_rows")(shp) == -1
    shp = _native.ConvShape(256, 80, 32, 64, 64, 3, 1)
    assert 0 < lib.raw("ds_conv_stats_rows")(shp) <= 256 * 80          # one row per M-tile


def test_python_surface_matches_reference_contract():
    from deepspeaker_pytorch_amd.model import DeepSpeakerModel, PairwiseDistance, TripletMarginLoss
    m = DeepSpeakerModel(512, 1211)
    sd = m.state_dict()
    assert len(sd) == 76
    assert sum(p.numel() for p in m.parameters()) == 12245371
    assert sd["model.conv1.weight"].shape == (64, 1, 5, 5)
    assert sd["model.layer4.0.conv2.weight"].shape == (512, 512, 3, 3)
    assert sd["model.fc.weight"].shape == (512, 2048) and sd["model.classifier.weight"].shape == (1211, 512)
    assert m.embedding_size == 512
    for attr in ("conv1", "bn1", "layer1", "conv4", "bn4", "layer4", "avgpool", "fc", "classifier", "relu"):
        assert hasattr(m.model, attr)
    # checkpoint round trip with an optimizer, as train_triplet.py:325-327 / :177-186 do
    opt = torch.optim.Adagrad(m.parameters(), lr=0.1, lr_decay=1e-4)
    ck = {"epoch": 1, "state_dict": m.state_dict(), "optimizer": opt.state_dict()}
    m2 = DeepSpeakerModel(512, 1211)
    m2.load_state_dict(ck["state_dict"])
    assert TripletMarginLoss(0.1).margin == 0.1 and PairwiseDistance(2).norm == 2
    assert PairwiseDistance(1).norm == 1                     # any positive norm, as the reference's constructor
    with pytest.raises(ValueError):
        PairwiseDistance(0)


def test_no_cpu_fallback():
    from deepspeaker_pytorch_amd.model import DeepSpeakerModel, PairwiseDistance
    m = DeepSpeakerModel(512, 4).eval()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 1, 160, 64))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        PairwiseDistance(2).forward(torch.zeros(2, 8), torch.zeros(2, 8))


def test_library_sources_do_not_allocate_or_synchronise():
    """caller-owned buffers, no hipMalloc / hipFree / memset / synchronise inside the library, and no
    emulator branches in product source (the host stand-ins live in tests/emul/)."""
    csrc = os.path.join(ROOT, "deepspeaker-pytorch_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        txt = re.sub(r"//[^\n]*", "", open(os.path.join(csrc, f)).read())
        for banned in ("hipMalloc", "hipFree", "hipMemset", "hipDeviceSynchronize", "hipStreamSynchronize", "DS_EMULATED"):
            assert banned not in txt, (f, banned)


def test_missing_library_fails_loudly(tmp_path):
    from deepspeaker_pytorch_amd import _native
    with pytest.raises(_native.DeepSpeakerHipError, match="no fallback"):
        _native.NativeLib(str(tmp_path / "libdeepspeaker_hip.so"))


def test_product_package_never_imports_oracle():
    pkg = os.path.join(ROOT, "deepspeaker-pytorch_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "deepspeaker_oracle" not in txt and "torch_restatement" not in txt, f
                assert "/root/reference" not in txt, f


def test_synthetic_state_dict_is_the_oracle_stream():
    """bench.py takes its parameters from the package helper; the bench-size golden fixture was made from the
    oracle's generator: the two must be the same arrays."""
    import deepspeaker_oracle as O
    from deepspeaker_pytorch_amd.synthetic import synthetic_state_dict
    for seed, ncls, ns in ((0, 1211, 4), (5, 7, 2)):
        a, b = O.make_state_dict(seed=seed, num_classes=ncls, n_stages=ns), synthetic_state_dict(seed, ncls, ns)
        assert set(a) == set(b)
        assert all(np.array_equal(a[k], b[k]) for k in a)


def test_refine_policy_sizes_slots_from_history():
    """mining.RefinePolicy (host bookkeeping of the fp16 near-tie refinement): a generous slot count until a few calls
    have been observed, then power-of-two slot counts from the recent near-tie counts, the whole batch once they pass
    half of it, decaying with the history window."""
    from deepspeaker_pytorch_amd.mining import REFINE_CAP_MIN, REFINE_CAP_START, RefinePolicy
    pol = RefinePolicy()
    assert pol.cap_for(256) == REFINE_CAP_START and pol.cap_for(8) == 8         # a batch smaller than the slots
    for _ in range(RefinePolicy.WARM):
        pol.observe(1)
    assert pol.cap_for(256) == REFINE_CAP_MIN
    for seen, want in ((2, 4), (3, 8), (16, 32), (17, 64), (40, 128), (70, 256), (300, 256)):
        pol.observe(seen)
        assert pol.cap_for(256) == want, (seen, pol.cap_for(256))
    assert pol.max_seen == 300
    for _ in range(RefinePolicy.HISTORY):                                       # the large counts age out
        pol.observe(1)
    assert pol.cap_for(256) == REFINE_CAP_MIN


def test_refine_policy_band_follows_the_observed_error():
    """The band of the near-tie refinement is measured, not a constant: never below the floor, BAND_SAFETY x the largest
    fp16 error of d_n - d_p any slot showed in the window, and it comes back down when large errors age out."""
    from deepspeaker_pytorch_amd.mining import BAND_SAFETY, BAND_WINDOW, REFINE_BAND, RefinePolicy
    pol = RefinePolicy()
    assert pol.band_for() == REFINE_BAND and pol.err_samples == 0
    pol.observe(1, 2e-4, 30, 4.1e-4)
    assert pol.band_for() == REFINE_BAND                               # 2.5 x 2e-4 is still under the floor
    assert pol.embedding_error_observed == 4.1e-4                      # the path's distance to the 1e-3 contract, as sampled
    pol.observe(0, None, 0)                                            # a whole-batch call sampled nothing
    assert pol.err_samples == 30 and len(pol.errs) == 1
    pol.observe(2, 3e-3, 4, 1.1e-3)                                    # weights that spread the embeddings apart
    assert pol.embedding_error_observed == 1.1e-3
    assert abs(pol.band_for() - BAND_SAFETY * 3e-3) < 1e-12 and pol.err_max_ever == 3e-3 and pol.err_samples == 34
    for _ in range(BAND_WINDOW):
        pol.observe(1, 1e-4, 4)
    assert pol.band_for() == REFINE_BAND and pol.err_max_ever == 3e-3  # aged out of the window, remembered in the record


def test_batch_counters_survive_reassignment_of_any_module():
    """`_bump_batches_tracked` keeps the twelve BatchNorm counters as views of one tensor; a counter that was re-assigned
    (load_state_dict(assign=True), manual replacement) must be noticed whichever module it belongs to."""
    import torch
    from deepspeaker_pytorch_amd.model import DeepSpeakerModel
    m = DeepSpeakerModel(512, 4)
    m._bump_batches_tracked(1)
    mods = m._bn_modules()
    assert all(int(b.num_batches_tracked) == 1 for b in mods)
    mods[5]._buffers["num_batches_tracked"] = torch.tensor(7, dtype=torch.int64)      # a MIDDLE module
    m._bump_batches_tracked(3)
    assert [int(b.num_batches_tracked) for b in mods] == [4] * 5 + [10] + [4] * 6
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m.load_state_dict(sd, assign=True)
    m._bump_batches_tracked(1)
    assert [int(b.num_batches_tracked) for b in m._bn_modules()] == [5] * 5 + [11] + [5] * 6


def test_weight_init_statistics_match_the_reference_rule():
    """a14 (reference model.py:114-120): conv weights ~ N(0, sqrt(2 / (k*k*out_channels))), BatchNorm weight 1 / bias 0.
    The sample standard deviation of each filter bank must sit within 5 standard errors of the rule's."""
    import math
    import torch
    from deepspeaker_pytorch_amd.model import DeepSpeakerModel
    torch.manual_seed(123)
    m = DeepSpeakerModel(512, 10)
    checked = 0
    for name, p in m.named_parameters():
        if p.dim() == 4:
            co, _, kh, kw = p.shape
            std = math.sqrt(2.0 / (kh * kw * co))
            n = p.numel()
            got = float(p.detach().double().std())
            assert abs(got - std) < 5 * std / math.sqrt(2 * n), (name, got, std)
            assert abs(float(p.detach().double().mean())) < 5 * std / math.sqrt(n), name
            checked += 1
        elif ".bn" in name and name.endswith("weight"):
            assert bool((p == 1).all()), name
        elif ".bn" in name and name.endswith("bias"):
            assert bool((p == 0).all()), name
    assert checked == 12


def test_replayed_pmc_traffic_belongs_to_this_build():
    """bench.py replays roofline.traffic from profiles/pmc_traffic.json (HBM counters cannot be read from inside the
    process).  The entry of the headline family records the digest of the kernel sources it was collected on; a kernel
    change without a new PMC pass (tools/pmc_run.sh, tools/pmc_summary.py, tools/pmc_traffic_update.py) fails HERE, and
    bench.py prints traffic = null with the reason instead of a stale figure."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("bench_for_digest", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    entry = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["f16"]
    assert entry["kernel_sources_sha256"] == bench.conv_sources_digest("f16"), \
        "fp16 convolution sources changed since " + entry["source"] + ": collect the PMC passes again"
    assert os.path.exists(os.path.join(ROOT, entry["source"]))
