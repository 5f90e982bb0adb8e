"""bench.py's own launcher (no GPU needed): `python bench.py --gpus N` with no WORLD_SIZE spawns N ranks under
torch.distributed.run, rank 0 prints the one JSON line; too few devices is one clear message, not a launcher trace."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(kw)
    return env


def test_self_launch_two_ranks_gloo():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launch-selftest"], capture_output=True, text=True,
                         env=_env(DS_BENCH_BACKEND="gloo"), timeout=300)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]           # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["world_size"] == 2 and d["rank_sum"] == 3.0
    assert "torch.distributed.run" in out.stderr          # it said what it launched


def test_too_few_devices_is_one_clear_message():
    # this container has no GPU: the RCCL launch must refuse before spawning anything
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2"], capture_output=True, text=True,
                         env=_env(HIP_VISIBLE_DEVICES=""), timeout=300)
    assert out.returncode == 2
    assert "needs 2 devices" in out.stderr and "Traceback" not in out.stderr
    assert "torch.distributed.run" not in out.stderr


def test_launcher_world_size_mismatch_names_both():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--launch-selftest"], capture_output=True, text=True,
                         env=_env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), timeout=300)
    assert out.returncode != 0 and "WORLD_SIZE=2" in out.stderr and "--gpus 4" in out.stderr


def _run_emulated(*flags, nproc=2, timeout=900):
    """`torch.distributed.run --nproc-per-node N tests/emul_bench.py ...`: bench.py's main() as the driver launches it,
    kernels on the host emulator, gloo (tests/emul_bench.py)."""
    import socket
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "emul_bench.py"), "--gpus", str(nproc), *flags]
    from emul_util import emul_lib
    emul_lib()                                            # build the emulated library once, before the ranks race for it
    out = subprocess.run(cmd, capture_output=True, text=True,
                         env=_env(DS_BENCH_BACKEND="gloo", OMP_NUM_THREADS="1", DS_EMUL_BENCH_TRIPLETS="1", DS_EMUL_CUS="1"),
                         timeout=timeout)
    assert out.returncode == 0, out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]            # rank 0 only
    return json.loads(lines[0])


import pytest  # noqa: E402


@pytest.mark.parametrize("tprec", ["bf16x3"])      # (the fp16 step under data parallelism: tests/test_distributed_gloo.py)
def test_train_bench_two_ranks_end_to_end_on_the_emulator(tprec):
    """`bench.py --train --gpus 2` end to end -- rendezvous, DeepSpeakerModel.enable_data_parallel, the
    grouped training step with its 30 exchanges per step (12 BatchNorm layers x 2 directions + 5 gradient buckets + the
    logged loss), barrier + max over ranks, the one line -- with two real ranks (gloo; the eval line's self-launch test
    above only reaches the rendezvous)."""
    d = _run_emulated("--train", "--train-precision", tprec, "--steps", "1", "--warmup", "0", "--repeats", "0",
                      "--train-settle-seconds", "0")
    assert d["n_gpus"] == 2 and d["world_size"] == 2 and d["steps"] == 1 and d["scaling"] == "weak"
    assert d["unit"] == "utterances/s" and d["dtype"] == tprec and d["value"] > 0
    assert d["all_reduce_per_step"] == 2 * 12 + 5 + 1
    assert d["config"]["parallelism"] == "dp2" and d["config"]["grad_comm"] == "shared"
    assert d["config"]["grad_reduce"] == "allreduce"
