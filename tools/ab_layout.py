"""Same-process A/B on one box (box-to-box spread is larger than the effects measured here):

  1. the padded LDS strides of the fp16 convolution's pixel tile (ds_conv_f16_set_layout_padding) -- isolated forwards,
     per layer, alternating padded / unpadded;
  2. what each part of the step costs on top of the forward (filter, loss, near-tie refinement, search);
  3. per-launch timing: events bound to the launch (ds_launch_timing_arm) vs an event pair recorded around it;
  4. the refinement window (mining.RefineWindow).

Tried with this tool and dropped (r04): a side stream of the lowest priority (2.005 vs 2.014 ms: within the spread) and a
side stream confined to 8 / 16 / 32 / 64 CUs by a CU mask (7.4 / 7.8 / 4.0 / 3.4 ms per step instead of 2.06: the side
work becomes the critical path).

    python tools/ab_layout.py [--rounds 6]"""
import argparse
import os
import statistics
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=6)
    args = ap.parse_args()
    from deepspeaker_pytorch_amd import mining
    from deepspeaker_pytorch_amd.mining import mine_semihard_negatives, select_triplets
    from deepspeaker_pytorch_amd.model import DeepSpeakerModel, TripletMarginLoss, get_engine
    from deepspeaker_pytorch_amd.synthetic import synthetic_state_dict
    dev = torch.device("cuda", 0)
    sd = synthetic_state_dict(seed=0, num_classes=1211)
    model = DeepSpeakerModel(512, 1211, precision="f16")
    model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    model = model.to(dev).eval()
    g = torch.Generator(device="cpu").manual_seed(1234)
    data_all = torch.randn(768, 1, 160, 64, generator=g).to(dev)
    data = list(data_all.split(256))
    c1 = torch.randint(0, 64, (256,), generator=g)
    c2 = (c1 + 1 + torch.randint(0, 63, (256,), generator=g)) % 64
    c1, c2 = c1.to(dev), c2.to(dev)
    labels = torch.cat([c1, c1, c2])
    loss_fn = TripletMarginLoss(0.1)
    eng = get_engine()
    set_pad = eng.lib.raw("ds_conv_f16_set_layout_padding")

    def forwards(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.no_grad():
            for _ in range(n):
                model(data_all)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    ref = None
    res = {0: [], 1: []}
    layers = {0: {}, 1: {}}
    for on in (1, 0):
        set_pad(on)
        forwards(20)
        with torch.no_grad():
            e = model(data_all).clone()
        if ref is None:
            ref = e
        else:
            print("padded vs unpadded embeddings bit-identical:", bool(torch.equal(ref, e)))
    for r in range(args.rounds):
        for on in (1, 0):
            set_pad(on)
            forwards(3)
            res[on].append(forwards(20))
            eng.profile = []
            forwards(5)
            prof, eng.profile = eng.profile, None
            for label, fl, e0, e1, _ in prof:
                layers[on].setdefault(label, []).append(e0.elapsed_time(e1) * 1e3)
    set_pad(0)
    for on in (1, 0):
        print(f"layout padding {on}: forward ms median {statistics.median(res[on]):.4f}  all "
              + " ".join(f"{v:.3f}" for v in res[on]))
    for label in layers[1]:
        a, b = statistics.median(layers[1][label]), statistics.median(layers[0][label])
        print(f"  {label:34s} padded {a:7.1f} us   unpadded {b:7.1f} us   {100 * (b - a) / b:+.1f} %")

    # ---- side stream priority ----
    def step():
        with torch.no_grad():
            e_all = model(data_all)
            embs = list(e_all.split(256))
            sel = select_triplets(*embs, margin=0.1, model=model, inputs=data)
            loss = loss_fn.forward(*embs)
            mined = mine_semihard_negatives(embs[0], embs[1], c1, e_all, labels, side_stream=True)
        return loss, sel, mined

    def steps(n):
        keep = []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            keep.append(step())
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    normal = torch.cuda.Stream(device=dev)
    print(f"forwards only (no side stream work): {forwards(20):.4f} ms")

    # ---- plain embedding extraction with batches in flight (pipeline.BatchesInFlight) ----
    from deepspeaker_pytorch_amd.pipeline import BatchesInFlight
    for n in (1, 2, 3):
        pipe = BatchesInFlight(model, in_flight=n)
        ts = []
        for r in range(args.rounds):
            for _ in pipe([data_all] * 6):
                pass
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in pipe([data_all] * 40):
                pass
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / 40 * 1e3)
        print(f"embedding extraction, {n} batch(es) of 768 in flight: {statistics.median(ts):.4f} ms per batch "
              f"({768 / statistics.median(ts):.1f} k embeddings/s)")

    # ---- what each part of the step costs on top of the forward ----
    mining._side_streams[dev] = normal

    def variant(refine, loss, mine, mine_side=True):
        def one():
            with torch.no_grad():
                e_all = model(data_all)
                embs = list(e_all.split(256))
                sel = select_triplets(*embs, margin=0.1, model=model if refine else None, inputs=data)
                l_ = loss_fn.forward(*embs) if loss else None
                m_ = mine_semihard_negatives(embs[0], embs[1], c1, e_all, labels, side_stream=mine_side) if mine else None
            return sel, l_, m_

        def run(n):
            keep = []
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                keep.append(one())
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n * 1e3
        run(5)
        return statistics.median([run(20) for _ in range(5)])

    for name, cfg in (("forward + filter (no refinement)", (False, False, False)),
                      ("forward + filter + loss", (False, True, False)),
                      ("forward + filter + refinement", (True, False, False)),
                      ("forward + filter + search", (False, False, True)),
                      ("forward + filter + search on main", (False, False, True, False)),
                      ("whole step, search on main", (True, True, True, False)),
                      ("whole step", (True, True, True))):
        print(f"{name:36s} {variant(*cfg):.4f} ms")
    def prof_steps(n):
        keep = []
        torch.cuda.synchronize()
        eng.profile = []
        t0 = time.perf_counter()
        for _ in range(n):
            keep.append(step())
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n * 1e3
        prof, eng.profile = eng.profile, None
        conv = sum(e0.elapsed_time(e1) for _, _, e0, e1, _ in prof) / n
        return dt, conv

    mining._side_streams[dev] = normal
    rows = {"off": [], "launch-bound events": [], "event pairs": []}
    for r in range(args.rounds):
        rows["off"].append((steps(20), 0.0))
        eng.self_timed_launches = True
        prof_steps(3)
        rows["launch-bound events"].append(prof_steps(20))
        eng.self_timed_launches = False
        prof_steps(3)
        rows["event pairs"].append(prof_steps(20))
    eng.self_timed_launches = True
    for k, v in rows.items():
        print(f"per-launch timing {k:20s}: step ms median {statistics.median(a for a, _ in v):.4f}, timed convolutions "
              f"{statistics.median(b for _, b in v):.4f} ms/step")

    # ---- the batch's forward as concurrent parts on their own streams (per-utterance results do not depend on batching) ----
    from deepspeaker_pytorch_amd.mining import refine_policy
    refine_policy(model).window = 4
    part_streams = [torch.cuda.Stream(device=dev) for _ in range(4)]

    def parts_step(n_parts):
        cur = torch.cuda.current_stream(dev)
        xs = list(data_all.split(768 // n_parts))
        outs = []
        with torch.no_grad():
            for k, xk in enumerate(xs):
                st_ = part_streams[k]
                st_.wait_stream(cur)
                with torch.cuda.stream(st_):
                    outs.append(model(xk))
            for k in range(len(xs)):
                cur.wait_stream(part_streams[k])
                outs[k].record_stream(cur)
            e_all = torch.cat(outs)
            embs = list(e_all.split(256))
            sel = select_triplets(*embs, margin=0.1, model=model, inputs=data)
            loss = loss_fn.forward(*embs)
            mined = mine_semihard_negatives(embs[0], embs[1], c1, e_all, labels, side_stream=True)
        return loss, sel, mined, e_all

    def run_parts(n_parts, n):
        keep = []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            keep.append(parts_step(n_parts) if n_parts > 1 else step())
        refine_policy(model).flush()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    with torch.no_grad():
        whole = model(data_all).clone()
    for n_parts in (2, 3, 4):
        e_parts = parts_step(n_parts)[3]
        torch.cuda.synchronize()
        print(f"forward as {n_parts} concurrent parts: embeddings bit-identical to the single launch sequence:",
              bool(torch.equal(e_parts, whole)))
    res_p = {1: [], 2: [], 3: [], 4: []}
    for r in range(args.rounds):
        for n_parts in res_p:
            run_parts(n_parts, 5)
            res_p[n_parts].append(run_parts(n_parts, 20))
    for n_parts, v in res_p.items():
        print(f"whole step, forward as {n_parts} concurrent part(s): median {statistics.median(v):.4f} ms  all "
              + " ".join(f"{t:.3f}" for t in v))
    refine_policy(model).window = 1

    for w in (2, 4, 8):
        refine_policy(model).window = w
        print(f"whole step, refinement window {w:2d}      {variant(True, True, True):.4f} ms")
        print(f"  ... and the search on the main stream  {variant(True, True, True, False):.4f} ms")
        print(f"forward + filter + refinement, w {w:2d}  {variant(True, False, False):.4f} ms")
    refine_policy(model).flush()
    refine_policy(model).window = 1


if __name__ == "__main__":
    main()
