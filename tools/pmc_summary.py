#!/usr/bin/env python3
"""Per-dispatch summary of the rocprofv3 --pmc passes collected by tools/pmc_run.sh.
    python tools/pmc_summary.py gpurun_out/<tag> [kernel-substring] > profiles/<name>.md
GRBM_GUI_ACTIVE is reported summed over the 8 XCDs (it reads 8 x 2.4 GHz x duration)."""
import collections
import csv
import glob
import sys


def load(path):
    d = collections.defaultdict(dict)
    for r in csv.DictReader(open(path)):
        k = int(r["Dispatch_Id"])
        d[k]["name"] = r["Kernel_Name"]
        d[k]["grid"] = int(r["Grid_Size"])
        d[k]["wg"] = int(r["Workgroup_Size"])
        d[k]["lds"] = int(r["LDS_Block_Size"])
        d[k]["t"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        d[k][r["Counter_Name"]] = float(r["Counter_Value"])
    return d


def main():
    root = sys.argv[1]
    sub = sys.argv[2] if len(sys.argv) > 2 else "conv_mfma"
    last_n = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    sets = {}
    for s in ("sq1", "sq2", "fetch", "write", "tcc"):
        fs = glob.glob(f"{root}/{s}/*/*counter_collection.csv")
        if fs:
            sets[s] = [v for k, v in sorted(load(fs[0]).items()) if sub in v["name"]][-last_n:]
    print("| kernel | wgs | LDS B | us | clock GHz | MFMA busy | wave-cycles: wait_any / wait_inst / active | LDS conflict/active | FETCH KiB | WRITE KiB | L2 hit |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    n = len(sets["sq1"])
    for i in range(n):
        a, b = sets["sq1"][i], sets["sq2"][i]
        f = sets.get("fetch", [None] * n)[i]
        w = sets.get("write", [None] * n)[i]
        t = sets.get("tcc", [None] * n)[i]
        gui = b["GRBM_GUI_ACTIVE"] / 8
        clk = gui / (b["t"] / 1e9) / 1e9
        busy = a["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * gui * a["t"] / b["t"])
        wc = a["SQ_WAVE_CYCLES"]
        name = a["name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        print(f"| `{name}` | {a['grid']//a['wg']} | {a['lds']} | {a['t']/1e3:.0f} | {clk:.2f} | {busy:.3f} | "
              f"{a['SQ_WAIT_ANY']/wc:.2f} / {a['SQ_WAIT_INST_ANY']/wc:.2f} / {a['SQ_ACTIVE_INST_ANY']/wc:.2f} | "
              f"{b['SQ_LDS_BANK_CONFLICT']/max(b['SQ_LDS_IDX_ACTIVE'],1):.2f} | "
              f"{f['FETCH_SIZE']:.0f} | {w['WRITE_SIZE']:.0f} | "
              f"{t['TCC_HIT_sum']/(t['TCC_HIT_sum']+t['TCC_MISS_sum']):.2f} |")


if __name__ == "__main__":
    main()
