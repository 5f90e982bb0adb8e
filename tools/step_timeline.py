#!/usr/bin/env python3
"""One steady-state eval step of a rocprofv3 --kernel-trace database, launch by launch: start offset, stream, duration
and the gap to the previous launch of the same stream, for the main stream (delimited by conv1) and everything that ran
beside it.   python tools/step_timeline.py <results.db> [step-index-from-the-end, default 3]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name[:58]


def main():
    db = sqlite3.connect(sys.argv[1])
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    rows = db.execute("select k.start, k.end, k.stream_id, s.display_name from rocpd_kernel_dispatch k "
                      "join rocpd_info_kernel_symbol s on k.kernel_id = s.id order by k.start").fetchall()
    marks = [r for r in rows if "conv5x5s2_c1_bf16_kernel<true" in r[3]]
    if len(marks) < back + 2:
        raise SystemExit("too few steps in the trace")
    t0, t1 = marks[-back - 1][0], marks[-back][0]
    main_stream = marks[-back - 1][2]
    ks = [r for r in rows if r[1] > t0 and r[0] < t1]
    print(f"step of {(t1 - t0) / 1e3:.1f} us; main stream {main_stream}; {len(ks)} launches")
    last_end = {}
    busy = {}
    for s, e, sid, name in ks:
        gap = (s - last_end[sid]) / 1e3 if sid in last_end else float("nan")
        last_end[sid] = e
        busy[sid] = busy.get(sid, 0.0) + (e - s) / 1e3
        tag = "main" if sid == main_stream else f"s{sid}"
        print(f"{(s - t0) / 1e3:9.1f} us  {tag:5s} {(e - s) / 1e3:8.1f} us  gap {gap:7.1f}  {short(name)}")
    for sid, b in busy.items():
        print(f"stream {sid}: kernels {b:.1f} us")
    # main-stream idle time inside the step
    mk = [(s, e) for s, e, sid, _ in ks if sid == main_stream]
    idle = sum(max(0, mk[i + 1][0] - mk[i][1]) for i in range(len(mk) - 1)) / 1e3
    print(f"main stream: gaps between its launches {idle:.1f} us")


if __name__ == "__main__":
    main()
