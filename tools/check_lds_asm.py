#!/usr/bin/env python3
"""Static check of a gfx950 assembly listing (hipcc --cuda-device-only -S): no instruction may read (or overwrite) a vector
register that an LDS read has been ISSUED into but not yet WAITED for.

pfb_spec.hip issues its FIR window reads as `asm volatile("ds_read_b64 %0, ...", "=v"(w))` and waits for them in a later asm
statement (`s_waitcnt lgkmcnt(0)` with the registers as "+v" operands): the hand-placed reads run ahead of the packed FMAs
that hide them.  Between the two statements the compiler believes the value is defined -- were it to copy or spill the
register there (another -D variant, another compiler, more register pressure) it would move stale bits without any
diagnostic.  This walks every kernel of the listing and reports such a use; tests/test_asm_hazards.py runs it
on the shipped build, tools/pfb_variants.sh on every variant before it is timed.

LDS operations of a wave complete in order, so `s_waitcnt lgkmcnt(N)` retires all but the N youngest outstanding ones (scalar
memory loads share the counter and may return out of order: with any of them outstanding only lgkmcnt(0) is trusted)."""
import re
import sys

REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs(operand: str):
    out = set()
    for m in REG.finditer(operand):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def split_ops(rest: str):
    rest = rest.split(";")[0]
    return [o.strip() for o in rest.split(",") if o.strip()]


def check(text: str):
    problems, kernel = [], None
    pending = []            # [(registers an LDS read will fill, line number)] of every outstanding LDS operation, in issue order
    smem = 0
    n_reads = 0
    saved = {}              # label -> the pending list an unconditional branch to it carried
    unreachable = False     # behind an s_branch, up to the next label
    for ln, line in enumerate(text.splitlines(), 1):
        s = line.strip()
        is_label = bool(re.match(r"^[\w.$]+:\s*(;.*)?$", s))
        if not s or (s.startswith((";", ".", "//")) and not (is_label and s.startswith(".L"))):
            continue
        if is_label:
            lab = s.split(":")[0]
            if lab.startswith("_Z") or not lab.startswith((".L", "L")):
                kernel, pending, smem, saved, unreachable = lab, [], 0, {}, False
            elif unreachable:
                pending, unreachable = list(saved.get(lab, [])), False
            elif lab in saved:
                pending = pending + [e for e in saved[lab] if e not in pending]     # either way in: the union
            continue
        if unreachable:
            continue
        parts = s.split(None, 1)
        op = parts[0]
        ops = split_ops(parts[1]) if len(parts) > 1 else []
        if op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", s)
            if m:
                n = int(m.group(1))
                if n == 0:
                    pending, smem = [], 0
                elif smem == 0 and len(pending) > n:
                    pending = pending[len(pending) - n:]
            continue
        if op in ("s_endpgm",):
            pending, smem = [], 0
            unreachable = True
            continue
        if op == "s_branch" and ops:
            saved[ops[0]] = saved.get(ops[0], []) + [e for e in pending if e not in saved.get(ops[0], [])]
            unreachable = True
            continue
        if op.startswith(("s_load_", "s_buffer_load_", "s_store_", "s_buffer_store_", "s_dcache_")):
            smem += 1                   # scalar memory operations share the counter and complete out of order
            continue
        is_lds_read = op.startswith(("ds_read", "ds_bpermute", "ds_permute", "ds_swizzle", "ds_consume", "ds_append", "ds_ordered")) or \
            (op.startswith("ds_") and "_rtn" in op)
        no_dest = op.startswith(("ds_write", "ds_add", "ds_or", "ds_and", "ds_xor", "ds_max", "ds_min", "global_store", "buffer_store",
                                 "flat_store", "scratch_store", "s_", "v_cmp", "v_cmpx", "global_atomic", "buffer_atomic", "v_nop")) and not is_lds_read
        used = set()
        for i, o in enumerate(ops):
            if i == 0 and not no_dest:
                continue
            used |= regs(o)
        written = regs(ops[0]) if (ops and not no_dest) else set()
        outstanding = set().union(*[p[0] for p in pending]) if pending else set()
        bad_r, bad_w = used & outstanding, written & outstanding
        if bad_r or bad_w:
            problems.append((kernel, ln, s, sorted(bad_r), sorted(bad_w)))
        if is_lds_read and ops:
            pending.append((regs(ops[0]), ln))
            n_reads += 1
        elif op.startswith("ds_"):
            pending.append((set(), ln))         # LDS writes / atomics count in lgkmcnt too (in order with the reads)
    return problems, n_reads


def main():
    text = open(sys.argv[1]).read()
    problems, n_reads = check(text)
    for k, ln, s, r, w in problems[:40]:
        print("%s line %d: `%s` %s%s before its LDS read is waited for" % (
            k, ln, s, ("reads v%s " % r) if r else "", ("overwrites v%s " % w) if w else ""))
    print("%d LDS reads checked, %d violations" % (n_reads, len(problems)))
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main())
