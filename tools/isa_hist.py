#!/usr/bin/env python3
"""Instruction histogram of one kernel in a gfx950 assembly listing (hipcc --cuda-device-only -S).

    python tools/isa_hist.py /tmp/pfb.s 'pfb_channelizeILi40ELb1ELi0' [--split s_barrier]

Prints the opcode-class counts of the whole kernel and, with --split, of each stretch between two
occurrences of the split opcode (the channelizer's phases are separated by s_barrier)."""
import collections
import re
import sys


def classify(op: str) -> str:
    if op.startswith("v_pk_"):
        return "valu_pk"
    if op.startswith(("v_fma_", "v_fmac_", "v_mac_")):
        return "valu_fma"
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_"):
        return "valu_other"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, name = sys.argv[1], sys.argv[2]
    split = sys.argv[sys.argv.index("--split") + 1] if "--split" in sys.argv else None
    lines = open(path).read().splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(name) + r"\w*:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    body = []
    for l in lines[start + 1:end + 1]:
        s = l.strip()
        if not s or s.startswith((";", ".", "//")) or s.endswith(":"):
            continue
        body.append(s.split()[0])
    tot = collections.Counter(classify(o) for o in body)
    print(f"{name}: {len(body)} instructions")
    print("  total:", dict(sorted(tot.items())))
    ops = collections.Counter(body)
    print("  top opcodes:", ops.most_common(24))
    if split:
        seg, k = [], 0
        for o in body + [split]:
            if o.startswith(split):
                c = collections.Counter(classify(x) for x in seg)
                print(f"  segment {k}: {len(seg)} instr", dict(sorted(c.items())))
                seg, k = [], k + 1
            else:
                seg.append(o)


if __name__ == "__main__":
    main()
