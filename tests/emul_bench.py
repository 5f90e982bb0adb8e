"""TEST INFRASTRUCTURE: bench.py's main() with the kernels on the host emulator, CPU tensors and gloo -- so that the launch
sequence of `bench.py --train --gpus N` (rendezvous, data-parallel model, the step's collectives, barrier, max over ranks,
the one line of rank 0) runs end to end on a box without GPUs.  Launched like the driver launches bench.py:

    DS_BENCH_BACKEND=gloo python -m torch.distributed.run --nproc-per-node 2 ... tests/emul_bench.py --gpus 2 --train ...

Everything that makes this possible is done HERE, from outside the product: the engine is bound to the emulated library,
the package's "tensors must be on a ROCm device" check is disabled, the workload is shrunk to what the emulator computes in
seconds (bench.BATCH_TRIPLETS / FRAMES / PRE_STEPS), bench.DEVICE_OVERRIDE = "cpu".  Numbers printed by such a run mean
nothing; its structure (keys, exchange counts, world size) is what tests/test_bench_launch.py asserts."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

torch.set_num_threads(1)
from emul_util import emul_lib  # noqa: E402
from deepspeaker_pytorch_amd import mining, model as M, optim  # noqa: E402
from deepspeaker_pytorch_amd.engine import Engine  # noqa: E402

M._engine = Engine(emul_lib())
for mod in (M, mining, optim):
    if hasattr(mod, "_require_cuda"):
        mod._require_cuda = lambda t, what: None

import bench  # noqa: E402

bench.DEVICE_OVERRIDE = "cpu"
bench.BATCH_TRIPLETS = int(os.environ.get("DS_EMUL_BENCH_TRIPLETS", "2"))
bench.FRAMES = int(os.environ.get("DS_EMUL_BENCH_FRAMES", "16"))
bench.PRE_STEPS = 0

if __name__ == "__main__":
    bench.main()
