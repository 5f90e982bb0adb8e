"""Data-parallel logic on CPU: world_size 2 and 4 over gloo (and the forced data-parallel launch sequence with one
rank), kernels on the host emulator.

Asserts the property SURVEY 8(e) asks for: an N-rank step (sharded triplets, all-reduced BatchNorm
statistics, all-gathered embeddings for cross-rank mining, all-reduced gradients) reproduces the
single-process step on the global batch -- loss, every parameter gradient, the running statistics and
the embeddings.  The RCCL path on real GPUs runs the same Python with backend "nccl"."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B_GLOBAL, T = 4, 16


def _setup_paths():
    for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)


def _problem(N_STAGES):
    _setup_paths()
    import deepspeaker_oracle as O
    sd = O.make_state_dict(seed=51, num_classes=4, n_stages=N_STAGES)
    xs = [O.make_input(seed=52 + i, batch=B_GLOBAL, frames=T) for i in range(3)]
    c1 = np.array([0, 1, 2, 0], np.int64)
    c2 = np.array([1, 2, 0, 3], np.int64)
    return sd, xs, c1, c2


def _run(rank, world, mine, reducer, N_STAGES=2, arith="f32"):
    _setup_paths()
    from emul_util import emul_lib
    from deepspeaker_pytorch_amd.distributed import triplet_train_step
    from deepspeaker_pytorch_amd.engine import BNParams, Engine
    sd, xs, c1, c2 = _problem(N_STAGES)
    eng = Engine(emul_lib())
    eng.lib.trace = {}
    tsd = {k: torch.from_numpy(np.array(v)) for k, v in sd.items()}
    if arith == "f16":
        pw = eng.pack_weights(tsd, N_STAGES, with_dgrad=True, with_f16=True, f32_banks=False, with_f16_dgrad=True)
    else:
        pw = eng.pack_weights(tsd, N_STAGES, with_dgrad=True)
    names = []
    for i in range(1, N_STAGES + 1):
        names += [f"model.bn{i}", f"model.layer{i}.0.bn1", f"model.layer{i}.0.bn2"]
    bns = {n: BNParams(tsd[n + ".weight"], tsd[n + ".bias"], tsd[n + ".running_mean"], tsd[n + ".running_var"])
           for n in names}
    n_loc = B_GLOBAL // world
    sl = slice(rank * n_loc, (rank + 1) * n_loc)
    xa, xp, xn = (torch.from_numpy(x[sl].copy()) for x in xs)
    labels = (torch.from_numpy(c1[sl].copy()), torch.from_numpy(c2[sl].copy()))
    res = triplet_train_step(eng, pw, bns, {n: b.weight for n, b in bns.items()}, xa, xp, xn, 0.1, reducer,
                             labels=labels, mine=mine, arith=arith)
    out = {"loss": res.loss.numpy()}
    for k, v in eng.lib.trace.items():
        out["calls/" + k] = np.array(v)
    eng.lib.trace = None
    if reducer is not None:
        out["n_all_reduce"] = np.array(reducer.n_all_reduce)
    for k, v in res.grads.items():
        out["grad/" + k] = v.numpy()
    for n, b in bns.items():
        out["rm/" + n] = b.running_mean.numpy().copy()
        out["rv/" + n] = b.running_var.numpy().copy()
    for nm, e in zip("apn", res.embeddings):
        out["emb_" + nm] = e.numpy()
    return out


def _worker(rank, world, port, mine, outdir, n_stages=2, force=False, grad_comm=None, grad_reduce=None, arith="f32"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _setup_paths()
    from deepspeaker_pytorch_amd.distributed import Reducer
    out = _run(rank, world, mine, Reducer(force=force, grad_comm=grad_comm, grad_reduce=grad_reduce), n_stages, arith)
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), **out)
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _check_against_single_process(tmp_path, world, mine, n_stages):
    ref = _run(0, 1, mine, None, n_stages)            # the whole batch in one process, no collectives
    ranks = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    n_loc = B_GLOBAL // world
    for r, got in enumerate(ranks):
        assert abs(float(got["loss"]) - float(ref["loss"])) <= 1e-5 * max(1.0, abs(float(ref["loss"])))
        for k, v in ref.items():
            if k.startswith("grad/"):
                err = np.linalg.norm(got[k] - v) / max(np.linalg.norm(v), 1e-30)
                assert err < 5e-5, (r, k, err)
            elif k.startswith(("rm/", "rv/")):
                np.testing.assert_allclose(got[k], v, rtol=1e-5, atol=1e-6, err_msg=k)
            elif k.startswith("emb_"):
                np.testing.assert_allclose(got[k], v[r * n_loc:(r + 1) * n_loc], rtol=2e-5, atol=2e-5)
    # collective-lean: per BatchNorm LAYER one statistics all-reduce forward and one backward (all three members
    # together), one per gradient bucket (stages + fc), one for the loss -- not one per BatchNorm call
    assert int(ranks[0]["n_all_reduce"]) == 2 * 3 * n_stages + (n_stages + 1) + 1
    # the data-parallel step is the SAME grouped launch sequence as the single-process step, split at the all-reduce:
    # one grouped reduce + one grouped apply per BatchNorm layer, no per-member fallback launches
    assert int(ranks[0]["calls/ds_bn_bwd_group_reduce_f32"]) == 3 * n_stages
    assert int(ranks[0]["calls/ds_bn_bwd_group_apply_f32"]) == 3 * n_stages
    assert "calls/ds_bn_bwd_reduce_f32" not in ranks[0].files and "calls/ds_bn_bwd_f32" not in ranks[0].files
    assert int(ref["calls/ds_bn_bwd_group_f32"]) == 3 * n_stages
    # all ranks hold identical global gradients
    for k in ranks[0].files:
        if k.startswith("grad/"):
            for other in ranks[1:]:
                np.testing.assert_array_equal(ranks[0][k], other[k])
    assert float(ref["loss"]) > 0


@pytest.mark.parametrize("mine", [False, True])
def test_two_ranks_reproduce_single_process(tmp_path, mine):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), mine, str(tmp_path)), nprocs=world, join=True)
    _check_against_single_process(tmp_path, world, mine, 2)


def test_four_ranks_full_model_reproduce_single_process(tmp_path):
    """world_size 4, the 4-stage network, one triplet per rank, with cross-rank mining"""
    world = 4
    mp.spawn(_worker, args=(world, _free_port(), True, str(tmp_path), 4), nprocs=world, join=True)
    _check_against_single_process(tmp_path, world, True, 4)


@pytest.mark.parametrize("mine", [False, True])
def test_forced_data_parallel_sequence_with_one_rank(tmp_path, mine):
    """Reducer(force=True) in a group of ONE takes every data-parallel branch (float64 sums -> all-reduce ->
    *_from_sums kernels, bucket all-reduces from inside the backward pass, all-gather / reduce-scatter of the mined
    candidates) and must reproduce the plain step: what `bench.py --train --force-collectives` and the GPU test run on
    RCCL."""
    mp.spawn(_worker, args=(1, _free_port(), mine, str(tmp_path), 2, True), nprocs=1, join=True)
    _check_against_single_process(tmp_path, 1, mine, 2)


@pytest.mark.parametrize("grad_comm,grad_reduce,mine", [("separate", "allreduce", False), ("shared", "rs_ag", False),
                                                        ("separate", "rs_ag", False), ("separate", "rs_ag", True)])
def test_gradient_exchange_variants_are_the_same_step(tmp_path, grad_comm, grad_reduce, mine):
    """Reducer(grad_comm=..., grad_reduce=...): the buckets on their own communicator (reduced from inside the backward
    pass) or on the shared one (after it), as one all-reduce or as reduce-scatter + all-gather (bucket sizes here are
    not multiples of the world size: the padded staging path) -- every combination is the single-process step, with
    the same number of exchanges."""
    world = 2
    # (mine=True: the overlapped bucket exchange on its own communicator next to the embedding all-gather and its
    # reduce-scatter adjoint on the default one -- the combination that has never run on more than one real GPU)
    mp.spawn(_worker, args=(world, _free_port(), mine, str(tmp_path), 2, False, grad_comm, grad_reduce), nprocs=world,
             join=True)
    _check_against_single_process(tmp_path, world, mine, 2)


def test_reducer_rejects_unknown_modes_and_defaults_to_the_safe_ones(tmp_path):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(_free_port())
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        _setup_paths()
        from deepspeaker_pytorch_amd.distributed import Reducer
        r = Reducer(force=True)
        assert r.grad_comm == "shared" and r.grad_reduce == "allreduce" and not r.overlap_gradients
        assert r.grad_group is r.group                      # no second communicator unless asked for
        with pytest.raises(ValueError):
            Reducer(grad_comm="both")
        with pytest.raises(ValueError):
            Reducer(grad_reduce="ring")
        g = dist.new_group(ranks=[0])
        r2 = Reducer(force=True, grad_group=g)              # a pre-built communicator is taken as is
        assert r2.grad_group is g and r2.overlap_gradients
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,mine", [(2, False), (2, True), (1, False)])
def test_fp16_step_data_parallel_reproduces_single_process(tmp_path, world, mine):
    """The opt-in fp16 training step under data parallelism (world 2, and the forced sequence with one rank): global
    BatchNorm sums in one float64 all-reduce per layer and direction, f32 gradient buckets -- the same exchange count as the
    f32-class step -- reproduce the single-process fp16 step on the global batch up to what a last-bit difference in a
    statistic does to fp16-rounded tensors."""
    mp.spawn(_worker, args=(world, _free_port(), mine, str(tmp_path), 2, world == 1, None, None, "f16"), nprocs=world, join=True)
    ref = _run(0, 1, mine, None, 2, "f16")
    ranks = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    n_loc = B_GLOBAL // world
    for r, got in enumerate(ranks):
        assert abs(float(got["loss"]) - float(ref["loss"])) <= 2e-3 * max(1.0, abs(float(ref["loss"])))
        for k, v in ref.items():
            if k.startswith("grad/"):
                err = np.linalg.norm(got[k] - v) / max(np.linalg.norm(v), 1e-30)
                assert err < 3e-2, (r, k, err)              # (tiny maps: one flipped fp16 rounding at a clip boundary)
            elif k.startswith(("rm/", "rv/")):
                np.testing.assert_allclose(got[k], v, rtol=2e-3, atol=2e-3, err_msg=k)
            elif k.startswith("emb_"):
                np.testing.assert_allclose(got[k], v[r * n_loc:(r + 1) * n_loc], rtol=5e-3, atol=5e-3)
    assert int(ranks[0]["n_all_reduce"]) == 2 * 3 * 2 + (2 + 1) + 1
    assert int(ranks[0]["calls/ds_bn_bwd_group_reduce_f16"]) == 3 * 2 and int(ranks[0]["calls/ds_bn_bwd_group_apply_f16"]) == 3 * 2
    assert "calls/ds_bn_bwd_group_f16" not in ranks[0].files and int(ref["calls/ds_bn_bwd_group_f16"]) == 3 * 2
    for k in ranks[0].files:                                # all ranks hold identical global gradients
        if k.startswith("grad/"):
            for other in ranks[1:]:
                np.testing.assert_array_equal(ranks[0][k], other[k])


def test_gradient_bucketing_names():
    _setup_paths()
    from deepspeaker_pytorch_amd.distributed import needs_allreduce
    assert needs_allreduce("model.conv3.weight") and needs_allreduce("model.layer2.0.conv1.weight")
    assert needs_allreduce("model.fc.weight") and needs_allreduce("model.fc.bias")
    assert not needs_allreduce("model.bn1.weight") and not needs_allreduce("model.layer4.0.bn2.bias")
