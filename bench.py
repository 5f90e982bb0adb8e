#!/usr/bin/env python3
"""Headline benchmark: embeddings/s of the Deep Speaker hot path on MI355X.

One step = BASELINE.json configs[1]: eval-mode forward of the full ResCNN (64/128/256/512) on 256
synthetic triplets -- anchor / positive / negative batches of 256 [1,160,64] fbank utterances, i.e.
768 embeddings -- followed by the triplet margin loss, the triplet filter (with the near-tie refinement
that keeps the fp16 forward's selection identical to the reference's) and the semi-hard negative search
(reference model.py:185-218, 27-33; train_triplet.py:251-262), inputs resident in HBM.  With --gpus N > 1
every rank runs the same per-GPU work on its own triplets (weak scaling) and the embeddings are
all-gathered over RCCL so every rank holds the global batch for mining.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

BATCH_TRIPLETS = 256
FRAMES = 160
PRE_STEPS = 30                              # untimed settle-in steps before the W contract warm-ups (see pre_steps)
# tests/emul_bench.py only: run main() on CPU tensors with the kernels on the host emulator under gloo (a test of the
# N > 1 launch sequence without GPUs).  Never set from the command line or the environment; the product package refuses
# CPU tensors on its own.
DEVICE_OVERRIDE = None
FWD_FLOPS_PER_EMB = 2 * 1153335296
PEAK_TFLOPS = {"f32": 157.3,                # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
               "bf16x3": 2500.0, "bf16": 2500.0,    # bf16 MFMA dense peak
               "f16": 2500.0}               # fp16 MFMA dense peak (same rate as bf16)
KERNEL_NAME = {"f32": "conv_mfma_f32_kernel (implicit-GEMM 3x3/5x5, all tile shapes)",
               "bf16x3": "conv_mfma_bf16_kernel<X3=true> (3 bf16 MFMAs per product: hi*hi + hi*lo + lo*hi)",
               "bf16": "conv_mfma_bf16_kernel<X3=false>",
               "f16": "conv_mfma_f16_pkernel + conv_block3x3_f16_kernel (persistent workgroups; one v_mfma_f32_32x32x16_f16 per "
                      "product, fp16 activations in HBM; the BasicBlocks of stages 1-2 as one kernel each)"}
ARITH = {"f32": "exact-f32 MFMA (v_mfma_f32_32x32x2_f32), f32 activations",
         "bf16x3": "split-operand bf16 MFMA: x = hi + lo (two bf16), product = hi*hi + hi*lo + lo*hi on "
                   "v_mfma_f32_32x32x16_bf16, f32 accumulate, f32 activations; embeddings 6e-6 from the reference",
         "bf16": "bf16 MFMA operands (v_mfma_f32_32x32x16_bf16, f32 accumulate), f32 activations; embeddings ~3e-3 "
                 "from the reference (OUTSIDE the 1e-3 contract)",
         "f16": "fp16 MFMA operands (v_mfma_f32_32x32x16_f16, one MFMA per product, f32 accumulate), fp16 activations "
                "in HBM, f32 conv1 input / pooling / projection / loss; embeddings 3.7e-4 from the reference (contract "
                "1e-3, tests/test_gpu_bench_size.py); triplets within the MEASURED band (>= 1.25e-3) of the filter's decision "
                "boundary are re-embedded through the split-operand bf16 path inside the timed region (the near ties of "
                "`refine.window_steps` consecutive steps share one f32-class forward, the open window is flushed before the "
                "region ends; slots sized from the near-tie counts of earlier steps; more near ties than slots, or an "
                "observed fp16 error above half the band => the selection re-embeds the whole batch at f32-class precision "
                "when it is read), so the selection is the reference's; see `refine`"}


CONV_SOURCES = {      # the sources whose change invalidates a replayed PMC traffic figure (profiles/pmc_traffic.json)
    "f16": ("conv_mfma_f16.hip", "conv_mfma_f16_kernel.h", "conv_mfma_f16_pkernel.h", "conv_block_f16.hip", "ds_device.h"),
    "bf16x3": ("conv_mfma_bf16.hip", "conv_mfma_bf16_kernel.h", "ds_device.h"),
    "f32": ("conv_mfma_f32.hip", "ds_device.h"),
}


def conv_sources_digest(precision):
    """sha256 over the kernel sources of `precision`'s convolution family (the GPU box has no .git: a content hash, not
    a commit).  profiles/pmc_traffic.json records the digest of the build its counters were collected on;
    tools/pmc_traffic_update.py writes it."""
    import hashlib
    h = hashlib.sha256()
    for name in CONV_SOURCES.get(precision, ()):
        with open(os.path.join(ROOT, "deepspeaker-pytorch_amd", "csrc", name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def cpu_baseline(sd_np, budget_s=7.0):
    """The reference's CPU forward, eval mode, fp32, on the host cores, at B = 32 and B = 256.

    kind "reference": the UNMODIFIED /root/reference/model.py (`DeepSpeakerModel(512, 1211).eval()`, BASELINE.md 4
    steps 2-3) when that tree exists on this host; kind "port": its torch-ATen restatement
    (oracle/torch_restatement.py, pinned to the reference's recorded outputs by tests/test_oracle_golden.py) where it does
    not -- the GPU box has no /root/reference.  Both run the same oneDNN convolutions.

    Two figures: `all_cores` -- torch.set_num_threads(os.cpu_count()), the plan's protocol -- and `value`, the best of a
    short thread-count scan that includes it (oneDNN degrades badly when over-subscribed; `cores` = the threads of the
    reported value).  Bounded samples."""
    ncpu = os.cpu_count() or 1
    sd = {k: torch.from_numpy(np.array(v)) for k, v in sd_np.items()}
    kind, fwd = "port", None
    ref_dir = "/root/reference"
    if os.path.isfile(os.path.join(ref_dir, "model.py")):
        try:
            import importlib.util
            import warnings
            spec = importlib.util.spec_from_file_location("_reference_model", os.path.join(ref_dir, "model.py"))
            ref_mod = importlib.util.module_from_spec(spec)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                spec.loader.exec_module(ref_mod)
                torch.manual_seed(0)
                ref_model = ref_mod.DeepSpeakerModel(512, 1211).eval()
            ref_model.load_state_dict(sd)
            kind, fwd = "reference", ref_model.forward
        except Exception as exc:                    # (the port below is pinned to the same outputs)
            print(f"[bench] the reference model did not import ({exc}); timing its restatement", file=sys.stderr)
    if fwd is None:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))        # the checker, used by this leg only
        import torch_restatement as TR
        fwd = lambda x: TR.forward_eval(sd, x)      # noqa: E731
    x32 = torch.randn(32, 1, FRAMES, 64)
    scan = {}
    with torch.no_grad():
        for nt in sorted({min(ncpu, c) for c in (8, 16, 32, 64, 128)} | {ncpu}):
            torch.set_num_threads(nt)
            fwd(x32)                                              # warm-up at this thread count
            t0 = time.perf_counter()
            fwd(x32)
            scan[nt] = time.perf_counter() - t0
        cores = min(scan, key=scan.get)

        def sample(B, nt, budget):
            torch.set_num_threads(nt)
            x = torch.randn(B, 1, FRAMES, 64)
            fwd(x)                                                # 1 warm-up
            n, reps, t0 = 0, 0, time.perf_counter()
            while True:
                fwd(x)
                n += B
                reps += 1
                dt = time.perf_counter() - t0
                if (dt > budget and reps >= 3) or reps >= 64:
                    break
            return n / dt, n, reps, dt
        v32, v256 = sample(32, cores, budget_s), sample(256, cores, budget_s)
        # all cores: sampled like the others, unless the scan already showed it to be pathological (256 SMT threads on
        # the GPU box's EPYC: 9.5 s per forward of 32 against 0.06 s at 16 threads) -- then the scan's one timed forward
        # is the figure; three more of them would add half a minute to the line for a number nobody should use
        if cores == ncpu:
            all32 = v32
        elif scan[ncpu] > 1.0:
            all32 = (32 / scan[ncpu], 32, 1, scan[ncpu])
        else:
            all32 = sample(32, ncpu, budget_s / 2)
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            cpu_model = next((l.split(":", 1)[1].strip() for l in f if l.startswith("model name")), "unknown")
    except OSError:
        pass
    return {"value": round(v32[0], 1), "unit": "embeddings/s", "cores": cores, "host_cores": ncpu, "kind": kind,
            "cpu_model": cpu_model,
            "value_b256": round(v256[0], 1),
            "all_cores": {"value": round(all32[0], 1), "cores": ncpu, "forwards": all32[2],
                          "what": "torch.set_num_threads(os.cpu_count()), B = 32 (BASELINE.md section 4 step 1)"},
            "thread_scan_ms_b32": {str(k): round(v * 1e3, 1) for k, v in sorted(scan.items())},
            "sample": f"B=32: {v32[1]} utterances [1,{FRAMES},64] ({v32[2]} forwards, {v32[3]:.1f} s); B=256: {v256[1]} "
                      f"utterances ({v256[2]} forwards, {v256[3]:.1f} s); eval forward, fp32, torch {torch.__version__} CPU, "
                      f"{cores} threads (best of a scan over 8..{ncpu}; the host has {ncpu}); "
                      + ("the unmodified /root/reference/model.py" if kind == "reference" else
                         "oracle/torch_restatement.py (no /root/reference on this host)")}


def _backend():
    """"nccl" (= RCCL on ROCm); DS_BENCH_BACKEND=gloo only for the CPU test of the launch logic"""
    return os.environ.get("DS_BENCH_BACKEND", "nccl")


def self_launch(n_gpus, argv):
    """`python bench.py --gpus N` without a launcher: spawn N ranks of this script under torch.distributed.run (the
    command the driver would have used) and return its exit status.  Refuses up front when the host has fewer
    devices than ranks, so that the failure is one clear line and not N stack traces."""
    import socket
    import subprocess
    if _backend() == "nccl":
        have = torch.cuda.device_count()
        if have < n_gpus:
            print(f"bench.py --gpus {n_gpus}: needs {n_gpus} devices, this host shows {have} "
                  f"(HIP_VISIBLE_DEVICES={os.environ.get('HIP_VISIBLE_DEVICES', '<unset>')})", file=sys.stderr)
            return 2
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *argv]
    print("[bench] no launcher (WORLD_SIZE unset): " + " ".join(cmd), file=sys.stderr)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.run(cmd, env=env).returncode


def launch_selftest(rank, local_rank, world):
    """--launch-selftest: the launch / rendezvous / one-line contract without the workload (what the CPU test of
    the self-launcher runs under gloo, and a 10-second check of a new multi-GPU node under RCCL)."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    backend = _backend()
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        dist.init_process_group("nccl", device_id=dev)
    else:
        dev = torch.device("cpu")
        dist.init_process_group(backend)
    assert dist.get_world_size() == world and dist.get_rank() == rank
    t = torch.tensor([float(rank + 1)], dtype=torch.float64, device=dev)
    dist.all_reduce(t)
    dist.barrier()
    if rank == 0:
        print(json.dumps({"metric": "launch selftest", "n_gpus": world, "world_size": dist.get_world_size(),
                          "rank_sum": float(t.item()), "backend": backend}))
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--refine-window", type=int, default=1,
                    help="steps whose near ties share one refinement forward (mining.RefineWindow).  Default 1, the library's "
                         "default and what the reference loop gets: it reads every step's selection (train_triplet.py:262-264) "
                         "and a read closes the window.  Larger windows are reported as the `refine_window_8` secondary")
    ap.add_argument("--batches", type=int, default=4,
                    help="resident input batches the steps rotate through (each 768 utterances + labels, seeded): near-tie "
                         "counts, refinement slot sizing and the precision guard's samples then vary from step to step as "
                         "they do in a real loop")
    ap.add_argument("--profile-every", type=int, default=3,
                    help="the live roofline times the convolution launches of every N-th step of the timed region (a timed "
                         "launch costs ~5 us of completion-signal handling: all of them, 2 %% of the step).  Keep it coprime with "
                         "--refine-window: the steps a window's flush runs beside must be sampled at their true share "
                         "(every 4th step with a window of 8 sampled them at 40 %% instead of 12 %%: frac 0.375 instead of 0.405)")
    ap.add_argument("--refine-stream", default="side", choices=["side", "main"],
                    help="the near-tie refinement's f32-class forward on the side stream (beside the next forward) or on the "
                         "step's own stream (serialised)")
    ap.add_argument("--refine-slots", type=int, default=0,
                    help="smallest number of near-tie re-embedding slots of the fp16 path (0: the library default, "
                         "mining.REFINE_CAP_MIN); the policy grows them from observed counts either way")
    ap.add_argument("--ablate", default="", help="diagnosis only (the line is then NOT the contract's workload): comma list of "
                    "'refine' (filter without the near-tie re-embedding) and 'mine' (no semi-hard search)")
    ap.add_argument("--repeats", type=int, default=3,
                    help="after the contract's timed region (W warm-up + K steps -> `value`), time the same K-step region "
                         "this many more times and report median / min / max ms per step (box-to-box and DVFS spread)")
    ap.add_argument("--precision", default="f16", choices=["f32", "bf16x3", "bf16", "f16"],
                    help="arithmetic of the stage convolutions: fp16 operands + fp16 activations (default; 3.7e-4 from "
                         "the reference, near-tie selections refined at f32-class precision), split-operand bf16 MFMA "
                         "(f32-class accuracy, 6e-6), exact-f32 MFMA, or plain bf16 (~3e-3: outside the contract)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the untimed comparison runs (split-operand bf16 and exact f32) and the training step")
    ap.add_argument("--split-apn", action="store_true",
                    help="three separate 256-utterance forwards (the reference's call pattern, "
                         "train_triplet.py:215) instead of one 768-utterance forward")
    ap.add_argument("--streams", type=int, default=1,
                    help="steps in flight: consecutive (independent) steps alternate over this many HIP streams, so one "
                         "step's HBM- / latency-bound kernels run beside another's matrix kernels (the default 1 keeps "
                         "every kernel alone on the GPU, which is what the roofline object times)")
    ap.add_argument("--force-collectives", action="store_true",
                    help="take the N > 1 code path (RCCL all-gathers, barrier, max-over-ranks) with a single rank; "
                         "launch with torch.distributed.run --nproc-per-node 1 (self-test of the multi-GPU path)")
    ap.add_argument("--train", action="store_true",
                    help="time the TRAINING step instead (train-mode forward of the 768 utterances, triplet loss, "
                         "backward, gradient all-reduce, fused Adagrad): the step with collectives on its critical path")
    ap.add_argument("--train-precision", default="bf16x3", choices=["f32", "bf16x3", "f16"],
                    help="--train: arithmetic of the training step: the f32-class default (split-operand bf16, gradients 1e-4 "
                         "from the masked oracle), exact f32, or the OPT-IN fp16 step (fp16 activations and loss-scaled "
                         "gradients in HBM, one fp16 MFMA per product; embeddings / loss 1e-3, gradients 3e-3)")
    ap.add_argument("--graph", action="store_true",
                    help="--train: the whole step (forward of a / p / n, loss, backward, optimizer) captured into ONE HIP graph "
                         "and replayed per step (train_graph.GraphedTripletStep): no host work in the timed region")
    ap.add_argument("--train-settle-seconds", type=float, default=3.0,
                    help="--train: untimed steps for at least this long before the W warm-ups (reported as `settle_steps`): a "
                         "fresh process needs seconds of sustained load, not a step count, before its regions agree")
    ap.add_argument("--grad-comm", default=None, choices=["shared", "separate"],
                    help="--train: gradient buckets on the BatchNorm collectives' communicator, exchanged after the backward "
                         "pass (default; one program-ordered collective sequence per rank), or on a communicator of their "
                         "own, exchanged from inside the pass (overlapped; see distributed.Reducer)")
    ap.add_argument("--grad-reduce", default=None, choices=["allreduce", "rs_ag"],
                    help="--train: a bucket's exchange as one all-reduce or as reduce-scatter + all-gather")
    ap.add_argument("--pad-streams", type=int, default=0,
                    help="create this many idle HIP streams first (tuning probe: HIP deals streams to its hardware queues "
                         "round-robin in creation order, and which streams of a step share a queue changes its overlap)")
    ap.add_argument("--launch-selftest", action="store_true",
                    help="only rendezvous, one all-reduce and the one-line print (checks the launcher, not the kernels)")
    args = ap.parse_args()

    ablate = set(a for a in args.ablate.split(",") if a)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # no launcher around us: become the launcher (one rank per GPU), rank 0 of the children prints the line
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher set WORLD_SIZE={world}: pass --gpus {world} "
                         f"(or run `python bench.py --gpus {args.gpus}` without a launcher: it spawns its own ranks)")
    if args.launch_selftest:
        return launch_selftest(rank, local_rank, world)
    if DEVICE_OVERRIDE is not None:
        dev = torch.device(DEVICE_OVERRIDE)
    else:
        if torch.cuda.device_count() <= local_rank:
            raise SystemExit(f"rank {rank}: needs device {local_rank}, this host shows {torch.cuda.device_count()} GPU(s)")
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    dist = None
    multi = world > 1 or args.force_collectives           # the data-parallel code path
    if multi:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:                                  # --force-collectives without a launcher: a group of one
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if _backend() == "nccl":
            dist.init_process_group("nccl", device_id=dev)        # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(_backend())

    pad_streams = [torch.cuda.Stream(device=dev) for _ in range(max(0, args.pad_streams))]
    for st_ in pad_streams:
        with torch.cuda.stream(st_):
            torch.zeros(1, device=dev)
    from deepspeaker_pytorch_amd.mining import (REFINE_BAND, mine_semihard_negatives, refine_policy, select_triplets,
                                                side_stream as side_stream_of)
    from deepspeaker_pytorch_amd.model import DeepSpeakerModel, TripletMarginLoss, get_engine
    from deepspeaker_pytorch_amd.synthetic import synthetic_state_dict

    sd_np = synthetic_state_dict(seed=0, num_classes=1211)
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    loss_fn = TripletMarginLoss(0.1)
    eng = get_engine()
    # --batches resident batches, each anchors | positives | negatives as one [768,1,160,64] buffer in HBM with its
    # synthetic speaker ids (c1 = anchor/positive speaker, c2 = negative speaker, 64 speakers); step k takes batch
    # k % --batches.  Batch 0 is the batch of every earlier round's line (same generator, same draw order).
    n_batches = max(1, args.batches)
    batches = []
    for _ in range(n_batches):
        d_all = torch.randn(3 * BATCH_TRIPLETS, 1, FRAMES, 64, generator=g).to(dev)
        c1_ = torch.randint(0, 64, (BATCH_TRIPLETS,), generator=g)
        c2_ = (c1_ + 1 + torch.randint(0, 63, (BATCH_TRIPLETS,), generator=g)) % 64
        c1_, c2_ = c1_.to(dev), c2_.to(dev)
        batches.append({"all": d_all, "apn": list(d_all.split(BATCH_TRIPLETS)), "c1": c1_, "c2": c2_,
                        "labels": torch.cat([c1_, c1_, c2_])})
    data_all, data = batches[0]["all"], batches[0]["apn"]
    n_slots = max(2, args.streams)              # steps in flight, each with its own gather buffers (2: the `pipelined` secondary)
    # two sets of gather buffers per slot, used alternately: the search over one set (side stream, overlapped with the
    # next forward) is long done when that set is gathered into again two steps later
    emb_globs = [[torch.empty(world * 3 * BATCH_TRIPLETS, 512, device=dev) if multi else None for _ in range(2)]
                 for _ in range(n_slots)]
    lab_globs = [[torch.empty(world * 3 * BATCH_TRIPLETS, dtype=torch.int64, device=dev) if multi else None
                  for _ in range(2)] for _ in range(n_slots)]

    streams = [torch.cuda.Stream(device=dev) for _ in range(args.streams)] if args.streams > 1 else []

    def fence():
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        if multi:
            dist.barrier()
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)

    def load_model(precision, train_precision=None):
        model = DeepSpeakerModel(512, 1211, precision=precision, train_precision=train_precision)
        model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd_np.items()})
        return model.to(dev)

    def region(step, steps, finish=None, in_flight=None):
        """K steps bracketed by barrier + synchronize on both sides; max over ranks.  `finish`: enqueued after the K-th step,
        inside the region (closes what the steps left open: the refinement window).  `in_flight` (default --streams):
        consecutive steps alternate over this many HIP streams."""
        in_flight = args.streams if in_flight is None else in_flight
        fence()
        t0 = time.perf_counter()
        if in_flight > 1:
            cur = torch.cuda.current_stream(dev)
            for st_ in streams[:in_flight]:
                st_.wait_stream(cur)
            for i in range(steps):
                with torch.cuda.stream(streams[i % in_flight]):
                    step(i % in_flight)
            for st_ in streams[:in_flight]:
                cur.wait_stream(st_)
        else:
            for _ in range(steps):
                step()
        if finish is not None:
            finish()
        t_enq = time.perf_counter() - t0        # host time to enqueue the steps (the device may still be busy)
        fence()
        elapsed = time.perf_counter() - t0
        if multi:
            t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed, t_enq

    def pre_steps(warmup):
        """untimed settle-in steps run BEFORE the W contract warm-ups (reported as `pre_steps` in the line)"""
        return max(0, PRE_STEPS - warmup)

    def timed(step, steps, warmup, repeats=0, profile=True, finish=None, profile_every=1, settle_s=0.0):
        # profile: per-launch events around the convolutions of the timed region (the live roofline of the eval line); the
        # training steps are timed without them (a pair of events per launch is ~70 per fp16 training step -- measured
        # 12.7 ms per step with them, 9.6 without)
        def prof_on():
            eng.profile = [] if profile else None
            eng._profile_calls = {}
        prof_on()                               # warm-up with the event instrumentation on: the first
        for _ in range(pre_steps(warmup)):      # timing events of a process cost ~40 ms to create; and a fresh
            step()                              # box needs ~0.2 s of work before clocks / caches settle (setup,
        fence()                                 # not part of the W contract warm-up steps that follow)
        # the training legs: settle by TIME -- the first K-step region of a process measured 8 - 15 % slow however many
        # warm-up steps preceded it (r05 driver: 10.02 / 8.72 / 8.72 ms after 10 steps; r06: 20.7 / 18.25 / 18.05 after
        # 30): what has to pass is a couple of seconds of sustained load, not a step count
        n_settle = 0
        if settle_s > 0:
            t_settle = time.perf_counter()
            for _ in range(5):
                step()
            fence()
            dt = torch.tensor([time.perf_counter() - t_settle], dtype=torch.float64, device=dev)
            if multi:                           # every rank must run the same number of steps (they meet in collectives)
                dist.all_reduce(dt, op=dist.ReduceOp.MAX)
            batches = min(400, max(0, int(np.ceil(settle_s / max(float(dt.item()), 1e-3))) - 1))
            for _ in range(batches):
                for _ in range(5):
                    step()
            fence()
            n_settle = 5 * (1 + batches)
        extras["settle_steps"] = n_settle
        prof_on()
        for _ in range(warmup):
            step()
        for j, st_ in enumerate(streams[:args.streams] if args.streams > 1 else []):   # launch plans / allocator pools
            with torch.cuda.stream(st_):
                step(j)
        if finish is not None:
            finish()                            # nothing of the warm-up is left for the timed region to do
        prof_on()
        eng.profile_every = max(1, profile_every)
        elapsed, t_enq = region(step, steps, finish)
        eng.profile_every = 1
        if rank == 0:
            print(f"[bench] host enqueue {t_enq / steps * 1e3:.3f} ms/step, device-complete {elapsed / steps * 1e3:.3f} ms/step",
                  file=sys.stderr)
        prof, eng.profile = (eng.profile or []), None
        try:                                    # the launch-bound events of the region must all read back
            for p_ in prof:
                p_[2].elapsed_time(p_[3])
        except Exception as exc:                # (never seen; a driver that does not bind them must not cost the line)
            if rank == 0:
                print(f"[bench] launch-bound events failed ({exc}); the roofline is timed with event pairs over one more "
                      f"region -- `value` stays the first region's", file=sys.stderr)
            eng.self_timed_launches = False
            prof_on()
            region(step, steps, finish)
            prof, eng.profile = (eng.profile or []), None
        # the same region again, `repeats` times, without the per-launch events: the spread of the box
        again = [region(step, steps, finish)[0] / steps * 1e3 for _ in range(repeats)]
        if rank == 0 and again:
            print(f"[bench] repeats of the {steps}-step region, ms/step: " + ", ".join(f"{v:.3f}" for v in again), file=sys.stderr)
        return elapsed, prof, again

    def measure(precision, steps, warmup, repeats=0):
        model = load_model(precision).eval()
        if args.refine_slots > 0:
            refine_policy(model).cap_min = refine_policy(model).cap_start = args.refine_slots
        refine_policy(model).window = max(1, args.refine_window)
        last_mined = [[None, None] for _ in range(n_slots)]
        parity = [0] * n_slots
        sels = []
        step_no = [0]

        def step(slot=0):
            par = parity[slot]
            parity[slot] ^= 1
            bt = batches[step_no[0] % n_batches]            # the resident batches in rotation
            step_no[0] += 1
            data_all, data, c1, labels_loc = bt["all"], bt["apn"], bt["c1"], bt["labels"]
            emb_glob, lab_glob = emb_globs[slot][par], lab_globs[slot][par]
            if multi and last_mined[slot][par] is not None:
                last_mined[slot][par].wait()            # the search that read this buffer set two steps ago (done long since)
            with torch.no_grad():
                if args.split_apn:
                    embs = [model(x) for x in data]
                    e_all = torch.cat(embs)
                else:                                   # eval mode: per-utterance results do not depend on batching
                    e_all = model(data_all)
                    embs = list(e_all.split(BATCH_TRIPLETS))
                # cross-GPU semi-hard negative search over the all-gathered global batch (BASELINE configs[2]);
                # at N = 1 the candidate set is the local batch, so per-GPU work has the same shape.  The
                # gathers run on RCCL's stream while the local loss / filter kernels run on ours; the search (its
                # result feeds the NEXT batch) is enqueued on the side stream next to the near-tie refinement, so
                # this stream goes straight on to its next forward.  The timed region ends with a device-wide
                # synchronize: all of it is inside.
                if multi:
                    h_emb = dist.all_gather_into_tensor(emb_glob, e_all, async_op=True)
                    h_lab = dist.all_gather_into_tensor(lab_glob, labels_loc, async_op=True)
                # filter (train_triplet.py:251-262) with the near ties of the fp16 forward re-embedded at f32-class
                # precision; the loss call below re-uses the same distance pass
                sel = select_triplets(*embs, margin=0.1, model=None if "refine" in ablate else model, inputs=data,
                                      side_stream=args.refine_stream == "side")
                loss = loss_fn.forward(*embs)
                if "mine" in ablate:
                    mined = None
                elif multi:
                    with torch.cuda.stream(side_stream_of(dev)):    # the SIDE stream waits for the gathers: this one
                        h_emb.wait()                                # never stalls on xGMI
                        h_lab.wait()
                    mined = mine_semihard_negatives(embs[0], embs[1], c1, emb_glob, lab_glob, side_stream=True)
                else:
                    mined = mine_semihard_negatives(embs[0], embs[1], c1, e_all, labels_loc, side_stream=True)
            last_mined[slot][par] = mined
            sels.append(sel)
            if len(sels) > 4 * steps:
                del sels[:-steps]
            return loss, sel, mined

        elapsed, prof, again = timed(step, steps, warmup, repeats, finish=refine_policy(model).flush,
                                     profile_every=args.profile_every if precision == args.precision else 1)
        # The same kernels with NOTHING else on the chip: forwards only, back to back (no loss / filter / refinement /
        # search, hence no side stream).  Inside the step the near-tie refinement and the semi-hard search of step k run
        # on a side stream next to the forward of step k + 1 and take CUs from it, so the per-launch durations measured
        # above include that contention; this is the rate of the kernels themselves (reported as roofline.isolated,
        # never as `value`).
        isolated = None
        if repeats > 0 and not multi and not args.split_apn:
            with torch.no_grad():
                for _ in range(3):
                    model(data_all)
                fence()
                eng.profile = []
                t0 = time.perf_counter()
                for _ in range(10):
                    model(data_all)
                fence()
                iso_ms = (time.perf_counter() - t0) / 10 * 1e3
                iso_prof, eng.profile = eng.profile, None
            isolated = (iso_ms, iso_prof)
        refine = None
        if precision == "f16" and "refine" not in ablate:
            # what the near-tie refinement did in the steps just timed (read AFTER the timed regions: the step itself
            # never synchronises; an overflow would re-embed the whole batch when the selection is read)
            last = sels[-steps:]
            ties = [s_.n_near_ties for s_ in last]
            pol = model._refine_policy
            errs = [s_.observed_error[0] for s_ in last if s_.observed_error[0] is not None]
            fallbacks = sum(int(s_.refine_overflow or s_.band_exceeded) for s_ in last)
            refine = {"overflow_steps": sum(int(s_.refine_overflow) for s_ in last),
                      "band_violation_steps": sum(int(s_.band_exceeded) for s_ in last),
                      "embedding_error_observed": pol.embedding_error_observed,
                      "band": last[-1].band, "band_floor": REFINE_BAND, "band_observed_max": pol.err_max_window,
                      "band_observed_max_timed_steps": max(errs, default=None), "band_samples_total": pol.err_samples,
                      "band_violations_total": pol.band_violations,
                      "slots": [s_.amb_cap for s_ in last][-1], "window_steps": pol.window, "near_ties_mean": round(sum(ties) / len(ties), 2),
                      "near_ties_max": max(ties), "steps": len(last), "calls_total": pol.calls, "overflows_total": pol.overflows}
            if fallbacks:
                # a step whose near ties outnumbered its slots, or whose probes showed an error above half its band,
                # re-embeds the whole batch when its selection is READ (after the timed region): say so in the line
                refine["flag"] = (f"{fallbacks} of the {len(last)} timed steps fell back to a whole-batch f32-class "
                                  "re-embedding at read time; that work is NOT inside `value`")
        # The same step with TWO steps in flight: consecutive (independent) steps alternate over two HIP streams, so one
        # step's HBM- / latency-bound launches (conv1, pooling + projection, loss, filter, refinement, search) and the drain
        # of each persistent kernel run beside the other step's matrix kernels.  Reported as `pipelined`, never as `value`:
        # launches of two steps then share the chip, which the per-launch roofline of the contract line must not see.
        # Run LAST (after the training legs): the two extra streams shift which hardware queue every later stream of the
        # process lands on, and the training step's stream overlap is sensitive to that (measured: 18.5 -> 19.6 ms).
        if repeats > 0 and not multi and args.streams == 1 and not args.split_apn and precision == args.precision:
            def run_pipelined():
                while len(streams) < 2:
                    streams.append(torch.cuda.Stream(device=dev))
                for j in range(2):
                    with torch.cuda.stream(streams[j]):
                        step(j)
                        step(j)
                fence()
                refine_policy(model).flush()
                runs = [region(step, steps, refine_policy(model).flush, in_flight=2)[0] / steps * 1e3 for _ in range(3)]
                return {"steps_in_flight": 2, "ms_per_step": round(float(np.median(runs)), 3),
                        "value": round(emb_per_step / float(np.median(runs)) * 1e3, 1), "unit": "embeddings/s",
                        "runs_ms_per_step": [round(v, 3) for v in runs],
                        "what": "the same K-step region with consecutive steps alternating over two HIP streams "
                                "(every step complete inside the bracket); results are the same tensors"}
            extras["pipelined_fn"] = run_pipelined

            def run_window(win):
                # `value` is measured at the library's default window of 1: a caller that READS every selection
                # (train_triplet.py:262-264) closes the window every step and pays one refinement forward per step.  A loop
                # that defers its reads can share one refinement forward between `win` steps; the same region with that
                # window, reported next to `value` (never as `value`).
                pol = refine_policy(model)
                pol.flush()
                keep = pol.window
                pol.window = win
                for _ in range(2 * win):
                    step()
                pol.flush()
                runs = [region(step, steps, pol.flush)[0] / steps * 1e3 for _ in range(3)]
                pol.flush()
                pol.window = keep
                return {"window_steps": win, "ms_per_step": round(float(np.median(runs)), 3),
                        "value": round(emb_per_step / float(np.median(runs)) * 1e3, 1), "unit": "embeddings/s",
                        "runs_ms_per_step": [round(v, 3) for v in runs],
                        "what": f"the same K-step region with mining.RefinePolicy.window = {win}: the near ties of {win} "
                                "consecutive steps share one f32-class refinement forward (the open window is flushed inside "
                                "the region)" if win > 1 else
                                "the same K-step region with mining.RefinePolicy.window = 1, the library default: what a loop "
                                "that reads every step's selection pays"}
            if precision == "f16" and "refine" not in ablate:
                other = 8 if max(1, args.refine_window) == 1 else 1
                extras["window_fn"] = (other, run_window)
        if precision == args.precision and getattr(model, "f16_guard", None) is not None:
            # the fp16 path's precision guard (precision_guard.py): what it measured on this workload and which kernels
            # the timed forwards therefore ran (an escalation would make this line a bf16x3 line, and say so)
            extras["precision_guard"] = model.f16_guard.report()
        return elapsed, prof, again, refine, isolated

    red_modes = [None, None]
    extras = {}

    def measure_train(precision, steps, warmup, repeats=0):
        """The training step of the triplet regime (train_triplet.py:215-224): train-mode forwards of a / p / n
        (three BatchNorm statistic sets, as the reference), triplet loss, backward (gradient all-reduce inside),
        fused Adagrad (lr 0.1, lr_decay 1e-4: train_triplet.py:369-383)."""
        from deepspeaker_pytorch_amd.optim import create_optimizer
        # precision "f16": the opt-in fp16 step (fp16 activations / loss-scaled gradients in HBM, train_f16.py)
        model = (load_model("f16", "f16") if precision == "f16" else load_model(precision)).train()
        red = model.enable_data_parallel(force=args.force_collectives, grad_comm=args.grad_comm,
                                         grad_reduce=args.grad_reduce) if multi else None
        opt = create_optimizer(model, 0.1, "adagrad", lr_decay=1e-4)
        tstep_no = [0]

        def step(slot=0):
            # the three forwards of train_triplet.py:215 in lock-step over one batch (same values, three BatchNorm
            # statistic sets); under data parallelism: one statistics all-reduce per BatchNorm layer and direction,
            # gradient buckets all-reduced from inside the backward pass
            d_ = batches[tstep_no[0] % n_batches]["apn"]
            tstep_no[0] += 1
            out_a, out_p, out_n = model.forward_triplet(d_[0], d_[1], d_[2])
            loss = loss_fn.forward(out_a, out_p, out_n)
            if multi:
                loss = loss / world
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
            if multi:                           # the global loss the loop logs every step (train_triplet.py:226)
                red.all_reduce_sum_(loss.detach())
            return loss

        if args.graph:
            if multi:
                raise SystemExit("--graph: collectives inside a captured step are not supported")
            from deepspeaker_pytorch_amd.train_graph import GraphedTripletStep
            gstep = GraphedTripletStep(model, opt, margin=0.1, example=batches[0]["apn"])

            def step(slot=0):                   # noqa: F811  (the replay: three input copies + one graph launch)
                d_ = batches[tstep_no[0] % n_batches]["apn"]
                tstep_no[0] += 1
                return gstep(d_[0], d_[1], d_[2])
        elapsed, prof, again = timed(step, steps, warmup, repeats, profile=False, settle_s=args.train_settle_seconds)
        per_step = None
        if red is not None:
            red_modes[:] = [red.grad_comm, red.grad_reduce]
            n0 = red.n_all_reduce
            step()
            per_step = red.n_all_reduce - n0
            fence()
        return elapsed, prof, again, per_step

    def roofline_of(precision, prof, steps):
        # live roofline of the dominant kernel family (the implicit-GEMM convolution of `precision`): algorithmic
        # FLOPs of every launch / its event-measured duration on the launch stream.  (The refinement forward of the
        # fp16 path launches split-operand bf16 kernels on a handful of rows: those are not this kernel.)
        prof = [p for p in prof if p[4] == precision]
        flops = sum(p[1] for p in prof)
        ms = sum(p[2].elapsed_time(p[3]) for p in prof)
        by = {}
        for label, fl, e0, e1, _ in prof:
            d = by.setdefault(label, [0.0, 0.0, 0])
            d[0] += fl
            d[1] += e0.elapsed_time(e1)
            d[2] += 1
        achieved = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        peak = PEAK_TFLOPS[precision]
        # HBM bytes per launch are not measurable from inside the process: replayed from the committed PMC passes
        # (profiles/pmc_traffic.json, averaged over ALL convolution launches of one forward at this batch)
        traffic, traffic_src = None, None
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                t = json.load(f).get(precision)
            if t and (not args.split_apn) == (t["launch_batch"] == 3 * BATCH_TRIPLETS):
                have, want = t.get("kernel_sources_sha256"), conv_sources_digest(precision)
                if have is not None and have != want:
                    # the kernels changed after the counters were collected: the figure is not this build's
                    traffic_src = (f"STALE -- {t.get('source', 'profiles/pmc_traffic.json')} was collected on kernel sources "
                                   f"{have}, this build is {want}: re-run tools/pmc_run.sh + tools/pmc_traffic_update.py")
                    if rank == 0:
                        print("[bench] roofline.traffic: " + traffic_src, file=sys.stderr)
                else:
                    traffic = t["traffic_bytes_per_launch"]
                    traffic_src = (f"replayed from {t.get('source', 'profiles/pmc_traffic.json')} ({t.get('launches', '?')} launches"
                                   + (f", kernel sources {have}" if have else ", sources digest not recorded") + ")")
        except (OSError, ValueError, KeyError):
            pass
        r = {"bound": "mfma", "kernel": KERNEL_NAME[precision], "achieved": round(achieved, 2), "peak": peak,
             "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_source": traffic_src,
             "launches": len(prof), "avg_launch_ms": round(ms / max(len(prof), 1), 4), "steps_timed": steps,
             "conv_ms_per_step": round(ms / steps, 3),
             "by_layer_tflops": {k: round(v[0] / (v[1] * 1e-3) / 1e12, 1) for k, v in by.items() if v[1] > 0}}
        # the member of the family furthest below the roofline (what the next optimisation round goes after)
        worst = min(((k, v[0] / (v[1] * 1e-3) / 1e12) for k, v in by.items() if v[1] > 0), key=lambda kv: kv[1], default=None)
        if worst:
            r["worst_kernel"] = {"name": worst[0], "tflops": round(worst[1], 1), "frac": round(worst[1] / peak, 4)}
        # what this chip's matrix cores deliver when they do nothing else (tools/mfma_peak.hip, profiles/r02_mfma_peak.txt:
        # back-to-back MFMAs from registers with random operand bits; the nominal peak assumes 2.4 GHz sustained, the
        # chip's power management does not) -- context for `frac`, replayed, not measured in this run
        attainable = {"f16": 1680.0, "bf16x3": 1823.0, "bf16": 1823.0, "f32": 151.0}.get(precision)
        live = mfma_rate.get("bf16" if precision in ("bf16x3", "bf16") else precision)
        if live:                # measured in this process (measure_mfma_rate), on this box
            r["mfma_register_only_tflops"] = live["tflops"]
            r["mfma_register_only_source"] = live["what"]
            r["frac_of_register_only"] = round((3 if precision == "bf16x3" else 1) * achieved / live["tflops"], 4)
            real = mfma_rate.get("f16_real") if precision == "f16" else None
            if real:            # the same loop on the operand values this forward's kernels see
                r["mfma_register_only_real_operands_tflops"] = real["tflops"]
                r["mfma_register_only_real_operands_source"] = real["what"]
                r["frac_of_register_only_real_operands"] = round(achieved / real["tflops"], 4)
        elif attainable:        # (the f32 MFMA rate is not probed live: replayed from profiles/r02_mfma_peak.txt)
            r["mfma_register_only_tflops"] = attainable
            r["mfma_register_only_source"] = "replayed from profiles/r02_mfma_peak.txt"
            r["frac_of_register_only"] = round((3 if precision == "bf16x3" else 1) * achieved / attainable, 4)
        if precision == "bf16x3":
            # "achieved" counts each product once (algorithmic FLOPs); the matrix cores issue three MFMAs per
            # product, so their issue rate is 3x that
            r["mfma_issue_tflops"] = round(3 * achieved, 1)
            r["mfma_issue_frac"] = round(3 * achieved / peak, 4)
        return r

    mfma_rate = {}

    def measure_mfma_rate(kind):
        """What the matrix cores of THIS box deliver when they do nothing else (ds_mfma_rate_probe: independent 32x32x16
        MFMAs back to back from registers on every SIMD, random operand bits): six launches of ~6.5 ms timed with events,
        the median of the last three (the clock settles over the first).  The nominal peak assumes 2.4 GHz sustained; the
        chip clocks to its power budget."""
        import ctypes
        sink = torch.zeros(1, device=dev)
        flop = ctypes.c_double(0.0)
        st_ = eng._stream(sink)
        ts = []
        for _ in range(6):      # (launches of ~6.5 ms: shorter ones are still ramping -- 1.3 -> 1.58 PFLOP/s over eight 1.7 ms launches)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            eng.lib.call("ds_mfma_rate_probe", 1 if kind == "bf16" else 0, 40000, eng._p(sink), ctypes.byref(flop), st_)
            e1.record()
            torch.cuda.synchronize(dev)
            ts.append(e0.elapsed_time(e1))
        ms = float(np.median(ts[3:]))
        mfma_rate[kind] = {"tflops": round(flop.value / ms / 1e9, 1),
                           "what": f"measured in this run: ds_mfma_rate_probe, {kind} 32x32x16 MFMAs back to back from registers on "
                                   f"every SIMD, random operand bits, median of 3 launches of {ms:.2f} ms"}

    def measure_mfma_rate_real(model):
        """The same probe with operands taken from THIS forward's tensors: A fragments from the packed fp16 filter bank of the
        256-channel 3x3 layer, B fragments from the fp16 activation that layer reads (post clipped-ReLU).  Random operand
        bits are the worst case for the multiplier array's switching power; this is the rate on the values the kernels see."""
        import ctypes
        with torch.no_grad():
            taps = {}
            pw_, folded_ = model._packed(with_f16=True), model._folded()
            eng.forward_eval(batches[0]["all"][:64].contiguous(), pw_, folded_, taps=taps, precision="f16")
        act, bank = taps["stage3.b"], pw_.stages[2].l_conv2_f16
        zeros = float((act == 0).float().mean())
        sink = torch.zeros(1, device=dev)
        flop = ctypes.c_double(0.0)
        st_ = eng._stream(sink)
        ts = []
        for _ in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            eng.lib.call("ds_mfma_rate_probe_data", eng._p(bank), bank.numel(), eng._p(act), act.numel(), 40000, eng._p(sink),
                         ctypes.byref(flop), st_)
            e1.record()
            torch.cuda.synchronize(dev)
            ts.append(e0.elapsed_time(e1))
        ms = float(np.median(ts[3:]))
        mfma_rate["f16_real"] = {"tflops": round(flop.value / ms / 1e9, 1),
                                 "what": "measured in this run: ds_mfma_rate_probe_data, the same back-to-back f16 32x32x16 MFMA "
                                         "loop with A fragments read from the packed fp16 filter bank of the 256-channel 3x3 layer "
                                         f"and B fragments from the fp16 activation it reads ({zeros:.0%} zeros), median of 3 "
                                         f"launches of {ms:.2f} ms"}

    emb_per_step = 3 * BATCH_TRIPLETS * world
    if args.train:
        tprec = args.train_precision
        elapsed, prof, again, ar_per_step = measure_train(tprec, args.steps, args.warmup, args.repeats)
        if rank == 0:
            line = {
                "metric": "training utterances/sec (64-fbank x 160-frame utterances)",
                "value": round(emb_per_step * args.steps / elapsed, 1), "unit": "utterances/s", "n_gpus": world,
                "steps": args.steps, "warmup": args.warmup, "pre_steps": pre_steps(args.warmup),
                "settle_steps": extras.get("settle_steps", 0),
                "world_size": dist.get_world_size() if multi else 1,
                "ms_per_step": round(elapsed / args.steps * 1e3, 3),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": tprec, "data": "synthetic",
                "config": {"workload": "triplet-regime training step (train_triplet.py:215-224): train-mode forward of "
                                       "256 triplets = 768 x [1,160,64] utterances per GPU, triplet loss, backward, "
                                       "gradient all-reduce, fused Adagrad",
                           "batch_triplets": BATCH_TRIPLETS, "parallelism": f"dp{world}"}}
            tfl = emb_per_step * args.steps / elapsed * 3 * FWD_FLOPS_PER_EMB / 1e12
            line["algorithmic_tflops"] = round(tfl, 1)         # ~3x the forward's FLOPs per utterance
            line["frac_of_mfma_peak"] = round(tfl / (157.3 if tprec == "f32" else 2500.0), 4)
            if args.graph:
                line["graph"] = ("the whole step replayed from ONE HIP graph (train_graph.GraphedTripletStep): ~330 (fp16) / ~400 "
                                 "(f32-class) launches over two to four streams, filter re-pack and optimizer included")
            if tprec == "f16":      # train-mode embeddings 1.2e-3 - 1.3e-3, gradients 4e-3 - 6e-3: outside north_star's 1e-3
                line["outside_contract"] = True
            if ar_per_step is not None:
                # 12 BatchNorm layers x {forward, backward} + 5 gradient buckets + the logged loss
                line["all_reduce_per_step"] = ar_per_step
                line["config"]["grad_comm"], line["config"]["grad_reduce"] = red_modes
            if again:
                line["repeats_ms_per_step"] = {"median": round(float(np.median(again)), 3), "min": round(min(again), 3),
                                               "max": round(max(again), 3), "n": len(again)}
            # every K-step region of this run (the contract's first, then the repeats)
            line["regions_ms_per_step"] = [round(elapsed / args.steps * 1e3, 3)] + [round(v, 3) for v in again]
            print(json.dumps(line))
        if multi:
            dist.destroy_process_group()
        return

    elapsed, prof, again, refine, isolated = measure(args.precision, args.steps, args.warmup, args.repeats)
    if "window_fn" in extras:
        win_, fn_ = extras.pop("window_fn")
        extras[f"refine_window_{win_}"] = fn_(win_)
    if "pipelined_fn" in extras:
        extras["pipelined"] = extras.pop("pipelined_fn")()

    secondary = {}
    if world == 1 and not args.no_secondary:
        for prec in ("bf16x3", "f32"):                              # untimed comparisons: the f32-class paths
            if prec != args.precision:
                k2 = max(3, args.steps // 2)
                e2, p2, _, _, _ = measure(prec, k2, 2)
                secondary[prec] = (e2, p2, k2)
        # BASELINE configs[4]: variable-length inference (100-800 frames) + enrolment scoring, same arithmetic (before the
        # training steps: their working sets stay in the caching allocator)
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import varlen_bench
        # the models measured so far are done: their launch plans (each pins GBs of activation buffers) and the allocator's
        # cached blocks go first -- with them in place the variable-length run's own plans (2 GB each, 16 of them) evict and
        # re-allocate one another through the 12 GiB plan-cache bound, and the same code measured 27 k - 130 k utterances/s
        # from run to run inside this process while it measures 124 k standalone
        fence()
        eng.drop_eval_plans()
        torch.cuda.empty_cache()
        varlen = varlen_bench.run(load_model(args.precision).eval(), n_utt=4096, dev=dev)
        # the training legs: >= 20 steps per region after >= 10 warm-ups, the median of 3 regions (5 steps after 2
        # warm-ups could not tell a 20 % regression from box spread)
        # (each leg also settles for --train-settle-seconds of untimed steps first: see `timed`)
        kt, wt_, regions_t = max(20, args.steps), 10, 3
        # The training legs run in FRESH PROCESSES (`bench.py --train ...`): HIP deals streams to its 4 hardware queues
        # round-robin in creation order and two streams on one queue serialise, so inside this process the legs' stream
        # overlap depends on how many streams the eval part happened to create before them (measured, same box, same
        # code: 18.1 / 8.9 ms or 19.4 / 9.7 ms).  A fresh process has one alignment -- the one `--train` is measured in.
        def train_leg(tp, graph=False):
            import subprocess
            cmd = [sys.executable, os.path.abspath(__file__), "--train", "--train-precision", tp, "--steps", str(kt),
                   "--warmup", str(wt_), "--repeats", str(regions_t - 1), "--no-cpu-baseline"] + (["--graph"] if graph else [])
            try:
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
                lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
                if r.returncode == 0 and lines:
                    regs = json.loads(lines[-1])["regions_ms_per_step"]
                    return float(np.median(regs)) * 1e-3 * kt, "fresh process", regs
                print(f"[bench] training leg {tp} in a fresh process failed (rc {r.returncode}): {r.stderr[-300:]}", file=sys.stderr)
            except Exception as exc:            # (no subprocess: measure here after all)
                print(f"[bench] training leg {tp} in a fresh process failed: {exc}", file=sys.stderr)
            if graph:
                return None, "failed", []
            e_, _, again_, _ = measure_train(tp, kt, wt_, regions_t - 1)
            regs = [e_ / kt * 1e3] + list(again_)
            return float(np.median(regs)) * 1e-3 * kt, "this process", [round(v, 3) for v in regs]
        et, leg_where, regs_t = train_leg("bf16x3")
        et16, _, regs_t16 = train_leg("f16")
        etg, _, regs_tg = train_leg("bf16x3", graph=True)
        etg16, _, regs_tg16 = train_leg("f16", graph=True)
    # the arithmetic the timed forwards really ran in: the requested one, unless the fp16 guard escalated
    eff_prec = extras.get("precision_guard", {}).get("verdict", args.precision)
    if rank == 0 and dev.type == "cuda" and not args.no_secondary:
        try:                # (after every timed region: the probe heats the chip)
            for kind in ("f16", "bf16"):
                measure_mfma_rate(kind)
            if eff_prec == "f16":
                measure_mfma_rate_real(load_model("f16").eval())
        except Exception as exc:
            print(f"[bench] ds_mfma_rate_probe failed ({exc}); frac_of_register_only uses the replayed figure", file=sys.stderr)
    if rank == 0:
        value = emb_per_step * args.steps / elapsed
        # the launches of every --profile-every-th forward of the timed region carry timing events
        fwd_calls = 3 if args.split_apn else 1
        profiled_steps = -(-(args.steps * fwd_calls) // max(1, args.profile_every)) / fwd_calls
        out = {
            "metric": "embeddings/sec (64-fbank x 160-frame utterances)",
            "value": round(value, 1), "unit": "embeddings/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "pre_steps": pre_steps(args.warmup),
            "world_size": dist.get_world_size() if multi else 1, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": eff_prec,
            "data": "synthetic" + (" (ABLATED: " + ",".join(sorted(ablate)) + " -- not the contract workload)" if ablate else ""),
            "config": {"workload": "BASELINE configs[1]: full DeepSpeaker ResCNN (64/128/256/512) eval forward + "
                                   "triplet loss + filter (near ties refined) + semi-hard negative search over the "
                                   "(all-gathered) batch, 256 triplets = 768 x [1,160,64] utterances per GPU per step",
                       "batch_triplets": BATCH_TRIPLETS, "utterances_per_step_per_gpu": 3 * BATCH_TRIPLETS,
                       "frames": FRAMES, "parallelism": f"dp{world}",
                       "forward_calls_per_step": 3 if args.split_apn else 1, "steps_in_flight": max(1, args.streams),
                       "refine_window_steps": max(1, args.refine_window),
                       "resident_batches": f"{n_batches} seeded batches (inputs + speaker ids) in rotation, step k takes batch k % {n_batches}",
                       "arith": ARITH[eff_prec]},
            "roofline": roofline_of(eff_prec, prof, profiled_steps),
            "whole_forward_tflops": round(value * FWD_FLOPS_PER_EMB / 1e12, 2),
        }
        # the whole step (conv1, tail, loss, filter, refinement, search and launch gaps included) against the same peak
        out["roofline"]["step_frac"] = round(value * FWD_FLOPS_PER_EMB / 1e12 / PEAK_TFLOPS[eff_prec], 4)
        if "precision_guard" in extras:
            out["precision_guard"] = extras["precision_guard"]
        if again:       # the contract's region is `value`; these are the same region timed again (box spread, DVFS)
            out["repeats_ms_per_step"] = {"median": round(float(np.median(again)), 3), "min": round(min(again), 3),
                                          "max": round(max(again), 3), "n": len(again)}
        if refine is not None:
            out["refine"] = refine
        for k_ in ("refine_window_1", "refine_window_8"):
            if k_ in extras:
                out[k_] = extras[k_]
        if "pipelined" in extras:
            out["pipelined"] = extras["pipelined"]
        if isolated is not None:
            iso_ms, iso_prof = isolated
            ir = roofline_of(eff_prec, iso_prof, 10)
            out["roofline"]["isolated"] = {
                "what": "the same launches with nothing else on the chip (10 forwards back to back, no side stream)",
                "forward_ms": round(iso_ms, 3), "achieved": ir["achieved"], "frac": ir["frac"],
                "conv_ms_per_forward": ir["conv_ms_per_step"], "by_layer_tflops": ir["by_layer_tflops"],
                "worst_kernel": ir.get("worst_kernel")}
        for prec, (e2, p2, k2) in secondary.items():
            out[prec + "_path"] = {"value": round(emb_per_step * k2 / e2, 1), "unit": "embeddings/s", "steps": k2,
                                   "roofline": roofline_of(prec, p2, k2)}
        if "precision_guard" in out and "bf16x3_path" in out and eff_prec == "f16":
            # `value` is the fp16 kernels' rate BECAUSE the guard measured this network inside its threshold.  On a network it
            # escalates -- measured: an SGD-trained one whose embeddings have spread, raw fp16 8.6e-4 -- every eval forward
            # runs the f32-class kernels, and the same step runs at this rate instead:
            out["precision_guard"]["value_if_escalated"] = out["bf16x3_path"]["value"]
            out["precision_guard"]["value_if_escalated_what"] = (
                "embeddings/s of this step when the guard escalates (a trained network whose fp16 error estimate passes "
                f"{out['precision_guard']['threshold']:g}): the split-operand bf16 path, measured in this run (`bf16x3_path`)")
        if world == 1 and not args.no_secondary:
            out["train_step"] = {"value": round(emb_per_step * kt / et, 1), "unit": "utterances/s", "steps": kt,
                                 "ms_per_step": round(et / kt * 1e3, 3), "dtype": "bf16x3",
                                 "algorithmic_tflops": round(emb_per_step * kt / et * 3 * FWD_FLOPS_PER_EMB / 1e12, 1),
                                 "frac_of_bf16_peak": round(emb_per_step * kt / et * 3 * FWD_FLOPS_PER_EMB / 1e12 / 2500.0, 4),
                                 "measured_in": leg_where, "warmup": wt_, "regions_ms_per_step": regs_t,
                                 "what": "train-mode forward of a/p/n (three BatchNorm statistic sets) + triplet loss + "
                                         "backward + fused Adagrad (train_triplet.py:215-224); ~3x the forward FLOPs"}
            out["train_step_f16"] = {"value": round(emb_per_step * kt / et16, 1), "unit": "utterances/s", "steps": kt,
                                     "ms_per_step": round(et16 / kt * 1e3, 3), "dtype": "f16",
                                     "warmup": wt_, "regions_ms_per_step": regs_t16,
                                     "algorithmic_tflops": round(emb_per_step * kt / et16 * 3 * FWD_FLOPS_PER_EMB / 1e12, 1),
                                     "frac_of_f16_peak": round(emb_per_step * kt / et16 * 3 * FWD_FLOPS_PER_EMB / 1e12 / 2500.0, 4),
                                     # north_star's contract is 1e-3 on embeddings and loss: this mode's TRAIN-mode embeddings
                                     # measure 1.2e-3 - 1.3e-3 (bar 2e-3), its gradients 4e-3 - 6e-3 (bar 8e-3)
                                     "outside_contract": True,
                                     "what": "the same step in the OPT-IN fp16 mode (DeepSpeakerModel(train_precision='f16')): fp16 "
                                             "activations and loss-scaled gradients in HBM, forward and data-gradient convolutions "
                                             "on the fp16 matrix-core kernels.  OUTSIDE north_star's 1e-3: loss 3.5e-4, train-mode "
                                             "embeddings 1.2e-3 - 1.3e-3 (bar 2e-3), gradients 4e-3 - 6e-3 vs the masked oracle "
                                             "(bar 8e-3; tests/test_gpu_train_f16.py); the default step (`train_step`) is the "
                                             "in-contract one"}
            for key_, e_, regs_, dt_ in (("train_step_graph", etg, regs_tg, "bf16x3"), ("train_step_f16_graph", etg16, regs_tg16, "f16")):
                if e_ is not None:
                    out[key_] = {"value": round(emb_per_step * kt / e_, 1), "unit": "utterances/s", "steps": kt,
                                 "ms_per_step": round(e_ / kt * 1e3, 3), "dtype": dt_, "regions_ms_per_step": regs_,
                                 "what": "the same step replayed from ONE HIP graph (train_graph.GraphedTripletStep: forward of "
                                         "a / p / n, loss, backward on its streams, filter re-pack, fused optimizer with a "
                                         "device-side step count); no host work in the timed region"}
                    if dt_ == "f16":
                        out[key_]["outside_contract"] = True
            out["varlen"] = varlen
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sd_np)
        # key order of the printed line: the contract's keys, then what a reader must not lose if the line's tail is cut
        # (what the refinement did in the timed steps -- overflows / band violations / the embedding error it observed --
        # the roofline, the CPU baseline), then the secondaries
        head = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "pre_steps", "world_size", "ms_per_step",
                "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "precision_guard", "refine", "roofline", "cpu_baseline",
                "config")
        out = {**{k: out[k] for k in head if k in out}, **{k: v for k, v in out.items() if k not in head}}
        print(json.dumps(out))
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
