#!/usr/bin/env python3
"""Where do a kernel's scratch (spill) accesses and AGPR shuttles sit relative to its MFMA streams?

    hipcc ... --cuda-device-only -S -o k.s kernel.hip ; python tools/isa_spills.py k.s [name-filter]

Per kernel: instruction count, MFMAs, scratch loads / stores with the index of the MFMA they follow (0 = before the first
MFMA, N = behind the last), v_accvgpr moves, s_waitcnt vmcnt(0) inside the MFMA range."""
import bisect
import re
import sys


def main():
    txt = open(sys.argv[1]).read()
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    for m in re.finditer(r"\n(_Z\w+):\s*; @\1\n(.*?)\n\s*s_endpgm", txt, re.S):
        name, body = m.group(1), m.group(2)
        if flt and flt not in name:
            continue
        lines = [l.strip() for l in body.split("\n") if l.strip() and not l.strip().startswith((";", "."))]
        mf = [j for j, l in enumerate(lines) if l.startswith("v_mfma")]
        if not mf:
            continue
        def where(pred):
            return [bisect.bisect(mf, j) for j, l in enumerate(lines) if pred(l)]
        ld = where(lambda l: l.startswith("scratch_load"))
        st = where(lambda l: l.startswith("scratch_store"))
        acc = where(lambda l: l.startswith(("v_accvgpr_write", "v_accvgpr_read")))
        vm0 = [p for p in where(lambda l: l.startswith("s_waitcnt") and "vmcnt(0)" in l) if 0 < p < len(mf)]
        inner = lambda ps: sum(1 for p in ps if 0 < p < len(mf))
        print(f"{name}\n  instructions {len(lines)}, MFMAs {len(mf)}, scratch loads {len(ld)} ({inner(ld)} between MFMAs) at {ld[:24]}, "
              f"stores {len(st)} ({inner(st)} between MFMAs) at {st[:24]}, accvgpr moves {len(acc)} ({inner(acc)} between MFMAs), "
              f"s_waitcnt vmcnt(0) between MFMAs: {len(vm0)}")


if __name__ == "__main__":
    main()
