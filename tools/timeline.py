#!/usr/bin/env python3
"""Stream-level timeline of a rocprofv3 --kernel-trace database (rocpd sqlite): per training step (delimited by the
optimizer kernel) the wall span, the time with >= 1 kernel running, the time with kernels of two streams running
together, and per stream the busy time by kernel family.   python tools/timeline.py <results.db> [delimiter-kernel]"""
import re
import sqlite3
import sys
from collections import defaultdict


def family(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"^\(anonymous namespace\)::", "", name)
    m = re.match(r"[A-Za-z_0-9:]+", name)
    base = m.group(0) if m else name
    if base.startswith("wgrad"):
        return "filter gradient"
    if base.startswith("conv"):
        return "convolution fwd / data gradient"
    if base.startswith("bn_") or base in ("colsum_kernel", "partial_sum_f64_kernel"):
        return "BatchNorm passes"
    if base.startswith("at::") or base.startswith("__amd"):
        return "torch / copies"
    return "other"


def union(iv):
    iv = sorted(iv)
    tot, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        tot += cur_e - cur_s
    return tot


def main():
    db = sqlite3.connect(sys.argv[1])
    delim = sys.argv[2] if len(sys.argv) > 2 else "adagrad_kernel"
    rows = db.execute("select k.start, k.end, k.stream_id, s.display_name from rocpd_kernel_dispatch k "
                      "join rocpd_info_kernel_symbol s on k.kernel_id = s.id order by k.start").fetchall()
    marks = [r[1] for r in rows if delim in r[3]]
    if len(marks) < 3:
        raise SystemExit("fewer than 3 delimiter kernels in the trace")
    steps = list(zip(marks[1:-1], marks[2:]))               # skip the first (warm-up) interval
    print(f"{len(steps)} steps delimited by {delim}")
    agg = defaultdict(float)
    per_stream = defaultdict(lambda: defaultdict(float))
    for t0, t1 in steps:
        ks = [r for r in rows if r[0] >= t0 and r[1] <= t1 + 1]
        by_stream = defaultdict(list)
        for s, e, sid, name in ks:
            by_stream[sid].append((s, e))
            per_stream[sid][family(name)] += (e - s) / 1e6
        agg["span"] += (t1 - t0) / 1e6
        agg["busy"] += union([(s, e) for s, e, _, _ in ks]) / 1e6
        agg["sum"] += sum(e - s for s, e, _, _ in ks) / 1e6
        ubs = {sid: union(v) for sid, v in by_stream.items()}
        agg["stream_union_sum"] += sum(ubs.values()) / 1e6
    n = len(steps)
    print(f"per step: span {agg['span'] / n:.2f} ms, some kernel running {agg['busy'] / n:.2f} ms, idle "
          f"{(agg['span'] - agg['busy']) / n:.2f} ms, kernel time summed {agg['sum'] / n:.2f} ms, two streams together "
          f"{(agg['stream_union_sum'] - agg['busy']) / n:.2f} ms")
    for sid, fams in per_stream.items():
        tot = sum(fams.values()) / n
        print(f"  stream {sid}: {tot:.2f} ms  " + ", ".join(f"{k} {v / n:.2f}" for k, v in sorted(fams.items(), key=lambda kv: -kv[1])))


if __name__ == "__main__":
    main()
