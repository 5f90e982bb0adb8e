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
