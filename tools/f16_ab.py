#!/usr/bin/env python3
"""Within-process A/B of fp16-forward kernel variants at the bench size (768 utterances): every variant is a copy of
the library built with extra -D flags; the variants' forwards run interleaved, round after round, and the per-layer
medians are printed side by side (methodology: cdna_hip_programming.md rule 24).

    python tools/f16_ab.py --build-only base: ring9:-DDS_F16_RING_K3=9      (build container: cross-compiles)
    gpurun -- python tools/f16_ab.py base: ring9:-DDS_F16_RING_K3=9 [--rounds 12]   (GPU box: loads build/ab/*.so)

A variant spec is name:flag,flag,...  (empty flag list = the stock build).  Results are checked bitwise against the
first variant (the kernels must stay bit-identical unless a variant says otherwise with a trailing '!')."""
import ctypes
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from deepspeaker_pytorch_amd._native import NativeLib
from deepspeaker_pytorch_amd.engine import BNParams, Engine
from deepspeaker_pytorch_amd.synthetic import synthetic_state_dict

CSRC = os.path.join(ROOT, "deepspeaker-pytorch_amd", "csrc")


AB_DIR = os.path.join(ROOT, "tools", "_ab")        # travels to the GPU box with the snapshot (objects are git-ignored)


def build(name, flags):
    out = os.path.join(AB_DIR, f"libds_ab_{name}.so")
    if os.environ.get("GRAFT_REPO_ROOT") or "--no-build" in sys.argv:      # on the GPU box: use what was built here
        assert os.path.exists(out), f"{out} missing: run tools/f16_ab.py --build-only ... in the build container first"
        return out
    objd = os.path.join(AB_DIR, f"obj_{name}")
    os.makedirs(objd, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))

    def cc(s):
        o = os.path.join(objd, os.path.basename(s)[:-4] + ".o")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mllvm",
                        "-pragma-unroll-threshold=1000000", "-Wno-pass-failed", f"-I{CSRC}", f"-I{ROOT}/include", *flags,
                        "-c", "-o", o, s], check=True)
        return o

    with ThreadPoolExecutor(16) as ex:
        objs = list(ex.map(cc, srcs))
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs], check=True)
    return out


def main():
    args = [a for a in sys.argv[1:] if a not in ("--build-only", "--no-build")]
    rounds = 12
    if "--rounds" in args:
        i = args.index("--rounds")
        rounds = int(args[i + 1])
        del args[i:i + 2]
    specs = []
    for a in args:
        name, _, fl = a.partition(":")
        loose = name.endswith("!")
        specs.append((name.rstrip("!"), [f for f in fl.split(",") if f], loose))
    if not specs:
        specs = [("base", [], False)]
    build_only = "--build-only" in sys.argv
    with ThreadPoolExecutor(len(specs)) as ex:
        paths = list(ex.map(lambda s: build(s[0], s[1]), specs))
    if build_only:
        print("\n".join(paths))
        return
    dev = torch.device("cuda", 0)
    sd_np = synthetic_state_dict(0, 1211)
    sd = {k: torch.from_numpy(np.array(v)).to(dev) for k, v in sd_np.items()}
    x = torch.randn(768, 1, 160, 64, generator=torch.Generator(device="cpu").manual_seed(1234)).to(dev)
    names = []
    for i in range(1, 5):
        names += [f"model.bn{i}", f"model.layer{i}.0.bn1", f"model.layer{i}.0.bn2"]
    engines, packs, folds = [], [], []
    for p in paths:
        eng = Engine(NativeLib(p))
        pw = eng.pack_weights(sd, 4, with_f16=True)
        bns = {n: BNParams(sd[n + ".weight"], sd[n + ".bias"], sd[n + ".running_mean"], sd[n + ".running_var"]) for n in names}
        folded = {n: eng.bn_fold(b) for n, b in bns.items()}
        engines.append(eng), packs.append(pw), folds.append(folded)
    outs = []
    with torch.no_grad():
        for eng, pw, fo in zip(engines, packs, folds):
            for _ in range(3):
                e = eng.forward_eval_planned(x, pw, fo, precision="f16")
            outs.append(e.clone())
        torch.cuda.synchronize()
        for (name, _, loose), e in zip(specs, outs):
            same = torch.equal(e, outs[0])
            print(f"{name}: embeddings {'bitwise equal to' if same else 'DIFFER from'} {specs[0][0]}"
                  + ("" if same else f" (max |d| {float((e - outs[0]).abs().max()):.3e})"))
            assert same or loose, name
        per = [dict() for _ in specs]
        tot = [[] for _ in specs]
        for r in range(rounds):
            for vi, (eng, pw, fo) in enumerate(zip(engines, packs, folds)):
                eng.profile = []
                t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0.record()
                eng.forward_eval_planned(x, pw, fo, precision="f16")
                t1.record()
                torch.cuda.synchronize()
                for label, fl, e0, e1, _ in eng.profile:
                    per[vi].setdefault(label, []).append((e0.elapsed_time(e1) * 1e3, fl))
                eng.profile = None
                tot[vi].append(t0.elapsed_time(t1) * 1e3)
        # whole forwards back to back WITHOUT per-launch events (launch gaps as they are in a real step)
        b2b = [[] for _ in specs]
        for r in range(max(4, rounds // 2)):
            for vi, (eng, pw, fo) in enumerate(zip(engines, packs, folds)):
                t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0.record()
                for _ in range(10):
                    eng.forward_eval_planned(x, pw, fo, precision="f16")
                t1.record()
                torch.cuda.synchronize()
                b2b[vi].append(t0.elapsed_time(t1) * 100.0)
    labels = list(per[0].keys())
    print(f"\nmedian us per launch over {rounds} interleaved rounds (min in brackets), TFLOP/s of the median")
    print("layer".ljust(30) + "".join(n.rjust(30) for n, _, _ in specs))
    for lb in labels:
        row = lb.ljust(30)
        for vi in range(len(specs)):
            ts = np.array([t for t, _ in per[vi][lb]])
            n_per = len(ts) // rounds
            ts = ts.reshape(rounds, n_per).sum(1)
            fl = sum(f for _, f in per[vi][lb][:n_per])
            row += f"{np.median(ts):9.1f} [{ts.min():7.1f}] {fl / np.median(ts) / 1e6:7.0f} TF".rjust(30)
        print(row)
    row = "whole forward (with events)".ljust(30)
    for vi in range(len(specs)):
        row += f"{np.median(tot[vi]):9.1f} [{min(tot[vi]):7.1f}]".rjust(30)
    print(row)
    row = "forward, 10 back to back, no events".ljust(30)
    for vi in range(len(specs)):
        row += f"{np.median(b2b[vi]):9.1f} [{min(b2b[vi]):7.1f}]".rjust(30)
    print(row)


if __name__ == "__main__":
    main()
