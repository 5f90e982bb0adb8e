#!/usr/bin/env python3
"""What a collective costs inside the data-parallel training step: from a rocprofv3 --kernel-trace database of
`bench.py --train --force-collectives`, for every RCCL kernel of the last complete step the idle time of the device
before it (no kernel of any stream running), its duration and the idle time after it.
    python tools/dp_gaps.py <results.db> [delimiter-kernel-substring]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    delim = sys.argv[2] if len(sys.argv) > 2 else "adagrad"
    rows = list(db.execute("select name, start, end, stream_id from kernels order by start"))
    ends = [i for i, r in enumerate(rows) if delim in r[0]]
    if len(ends) < 3:
        print("fewer than three steps in the trace")
        return
    lo, hi = ends[-3] + 1, ends[-2] + 1          # the last-but-one step (the last one may be the counting step)
    step = rows[lo:hi]
    span = step[-1][2] - step[0][1]
    busy_until = step[0][1]
    idle_total = 0
    recs = []
    for i, (name, s, e, st) in enumerate(step):
        gap = max(0, s - busy_until)
        idle_total += gap
        recs.append((name, s, e, st, gap))
        busy_until = max(busy_until, e)
    print(f"step span {span/1e3:.1f} us, {len(step)} kernels, device idle (no kernel running) {idle_total/1e3:.1f} us\n")
    print("| collective kernel | stream | idle before us | duration us | idle after us | next kernel |")
    print("|---|---|---|---|---|---|")
    tot_b = tot_d = tot_a = n = 0
    for i, (name, s, e, st, gap) in enumerate(recs):
        if "ccl" in name.lower():
            after = recs[i + 1][4] if i + 1 < len(recs) else 0
            nxt = recs[i + 1][0][:40] if i + 1 < len(recs) else ""
            print(f"| `{name[:50]}` | {st} | {gap/1e3:.1f} | {(e-s)/1e3:.1f} | {after/1e3:.1f} | `{nxt}` |")
            tot_b += gap; tot_d += e - s; tot_a += after; n += 1
    if n:
        print(f"\n{n} collectives: idle before {tot_b/1e3:.1f} us, duration {tot_d/1e3:.1f} us, idle after {tot_a/1e3:.1f} us "
              f"(sum {(tot_b+tot_d+tot_a)/1e3:.1f} us of the {span/1e3:.1f} us step)")
    big = sorted(recs, key=lambda r: -r[4])[:12]
    print("\nlargest idle gaps of the step (us before kernel):")
    for name, s, e, st, gap in big:
        print(f"  {gap/1e3:8.1f}  stream {st}  {name[:70]}")


if __name__ == "__main__":
    main()
