#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite result (`rocprofv3 --kernel-trace --stats -d DIR -- cmd` writes
DIR/<host>/<pid>_results.db on ROCm 7.2) into the per-kernel table committed under profiles/.

    python tools/rocpd_stats.py gpurun_out/prof/*/*_results.db > profiles/r01_kernel_stats.md
"""
import sqlite3
import sys


def main(paths):
    for path in paths:
        db = sqlite3.connect(path)
        rows = list(db.execute(
            "select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), "
            "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(grid_x), max(workgroup_x) "
            "from kernels group by name order by 3 desc"))
        tot = sum(r[2] for r in rows)
        print(f"## {path}\n")
        print("| kernel | calls | total ms | avg us | min us | max us | % | arch VGPR | acc VGPR | SGPR | LDS B | max grid (threads) | wg |")
        print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
        for r in rows:
            name = r[0].replace("(anonymous namespace)::", "").replace("|", "/")
            if len(name) > 100:
                name = name[:97] + "..."
            print(f"| `{name}` | {r[1]} | {r[2]/1e6:.3f} | {r[3]/1e3:.2f} | {r[4]/1e3:.2f} | {r[5]/1e3:.2f} | "
                  f"{100*r[2]/tot:.1f} | {r[6]} | {r[7]} | {r[8]} | {r[9]} | {r[10]} | {r[11]} |")
        print(f"\ntotal kernel time {tot/1e6:.3f} ms\n")


if __name__ == "__main__":
    main(sys.argv[1:])
