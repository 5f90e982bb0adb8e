#!/usr/bin/env python3
"""Write the fp16 entry of profiles/pmc_traffic.json (what bench.py replays as roofline.traffic) from a PMC summary made
by tools/pmc_summary.py, stamped with the digest of the kernel sources it was collected on (bench.conv_sources_digest):
bench.py refuses to replay a figure whose digest is not the build's, and tests/test_host_abi.py fails until it is redone.

    python tools/pmc_traffic_update.py profiles/<round>_pmc_summary.md [precision] [launches]

traffic = (2 x FETCH_SIZE + WRITE_SIZE) KiB x 1024 averaged over the last `launches` convolution rows of the summary
(one forward): FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 32-byte requests of wide reads as half)."""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src = sys.argv[1]
    precision = sys.argv[2] if len(sys.argv) > 2 else "f16"
    launches = int(sys.argv[3]) if len(sys.argv) > 3 else 9
    fam = {"f16": ("conv_mfma_f16", "conv_block3x3_f16"), "bf16x3": ("conv_mfma_bf16",), "f32": ("conv_mfma_f32",)}[precision]
    rows = []
    for line in open(src):
        cells = [c.strip() for c in line.strip().strip("|").split("|")]
        if len(cells) >= 11 and cells[0].startswith("`") and any(k in cells[0] for k in fam):
            rows.append((cells[0], float(cells[8]), float(cells[9])))
    # whole forwards only: a forward's convolution launches start with the stage-1 launch (the widest map); average over
    # every complete run of `launches` rows that starts there
    first = rows[[r[0] for r in rows].index(next(r[0] for r in rows if "block3x3_f16_kernel<2" in r[0] or precision != "f16"))][0]
    starts = [i for i, r in enumerate(rows) if r[0] == first and i + launches <= len(rows)
              and all(rows[j][0] != first for j in range(i + 1, i + launches))]
    assert starts, "no complete forward in the summary"
    per_fwd = [sum((2 * f + w) * 1024 for _, f, w in rows[i:i + launches]) / launches for i in starts]
    traffic = sum(per_fwd) / len(per_fwd)
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    d = json.load(open(path))
    e = d.setdefault(precision, {})
    e.update({"traffic_bytes_per_launch": int(round(traffic)), "launch_batch": 768, "launches": launches,
              "source": os.path.relpath(os.path.abspath(src), ROOT), "kernel_sources_sha256": bench.conv_sources_digest(precision)})
    json.dump(d, open(path, "w"), indent=1)
    print(f"{precision}: {traffic / 1e6:.1f} MB per launch over {launches} launches; sources {e['kernel_sources_sha256']}")


if __name__ == "__main__":
    main()
