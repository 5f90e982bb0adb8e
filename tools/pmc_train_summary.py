#!/usr/bin/env python3
"""Per-kernel PMC summary of the training step: the two SQ passes collected by

    rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \\
              SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU -d <root>/sq1 --output-format csv -- python tools/train_bench.py ...
    rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS \\
              SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d <root>/sq2 --output-format csv -- (same command)

one row per (kernel, grid) of the last step: duration, MFMA busy, wave-cycle split, LDS conflict rate.
python tools/pmc_train_summary.py <root> > profiles/<name>.md"""
import collections
import glob
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from pmc_summary import load


def main():
    root = sys.argv[1]
    A = load(glob.glob(root + "/sq1/*/*counter_collection.csv")[0])
    B = load(glob.glob(root + "/sq2/*/*counter_collection.csv")[0])
    rows = collections.OrderedDict()
    for k in sorted(A):
        a, b = A[k], B.get(k)
        if b is None or a["name"] != b["name"]:
            continue
        name = a["name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        if not any(s in name for s in ("wgrad_mfma", "conv_mfma", "conv5x5s2_c1", "bn_", "fc_", "adagrad")):
            continue
        gui = b["GRBM_GUI_ACTIVE"] / 8
        wc = a["SQ_WAVE_CYCLES"]
        rows[(name, a["grid"])] = (a["t"] / 1e3, a["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * gui * a["t"] / b["t"]),
                                   a["SQ_WAIT_ANY"] / wc, a["SQ_WAIT_INST_ANY"] / wc, a["SQ_ACTIVE_INST_ANY"] / wc,
                                   b["SQ_LDS_BANK_CONFLICT"] / max(b["SQ_LDS_IDX_ACTIVE"], 1), a["grid"] // a["wg"])
    print("| kernel | workgroups | us | MFMA busy | wave-cycles: wait_any / wait_inst / active | LDS conflict/active |")
    print("|---|---|---|---|---|---|")
    for (name, _), v in rows.items():
        print(f"| `{name}` | {v[6]} | {v[0]:.0f} | {v[1]:.3f} | {v[2]:.2f} / {v[3]:.2f} / {v[4]:.2f} | {v[5]:.2f} |")


if __name__ == "__main__":
    main()
