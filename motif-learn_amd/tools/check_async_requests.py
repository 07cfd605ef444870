#!/usr/bin/env python3
"""Static check of kernels that request operands through volatile asm (zk_sep_strip.hip: s_load / ds_read issued by hand,
waited for by hand).  The compiler takes an asm's outputs for written at once; the hardware writes them LATER.  Until the
s_waitcnt lgkmcnt(0) that covers a request, nothing else may write or read its destination registers -- in particular no
rematerialised s_load of a kernel argument into a register set the compiler believes dead (that race, a pointer overwritten by
late table data, was a memory-aperture fault in round 3).

  check_async_requests.py file.s [kernel-name-substring]      exit 1 and a listing if any hazard exists

Walks the control-flow graph of every kernel in the assembly (hipcc -S --cuda-device-only): the set of in-flight destination
registers is propagated along fall-through and branch edges to a fixed point."""
import re
import sys


def regs(tok):
    tok = tok.strip().rstrip(",")
    m = re.match(r"([sv])\[(\d+):(\d+)\]$", tok)
    if m:
        base = 0 if m.group(1) == "s" else 1000
        return {base + i for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.match(r"([sv])(\d+)$", tok)
    if m:
        return {(0 if m.group(1) == "s" else 1000) + int(m.group(2))}
    if tok == "vcc":
        return {106, 107}
    if tok == "exec":
        return {126, 127}
    return set()


NO_DEST = ("s_cmp", "s_cbranch", "s_branch", "s_waitcnt", "s_nop", "s_barrier", "s_endpgm", "global_store", "ds_write", "scratch_store",
           "s_bitcmp", "s_setprio", "buffer_store", "flat_store", "s_sleep", "s_setpc", "s_sendmsg", "s_trap", "v_cmpx")


def kernels(text):
    cur, name = None, None
    for line in text.splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name, cur = m.group(1), []
            continue
        if cur is not None:
            cur.append(line)
            if line.strip().startswith("s_endpgm"):
                yield name, cur
                cur = None


def check(name, lines):
    # instructions with their asm flag
    ins, labels, in_asm = [], {}, False
    for raw in lines:
        t = raw.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        m = re.match(r"^(\.LBB\w+):", t)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        if not t or t.startswith(";") or t.startswith("."):
            continue
        t = t.split(";")[0].strip()
        parts = re.split(r"[ ,\t]+", t)
        ins.append((parts[0], parts[1:], in_asm, t))
    n = len(ins)
    state_in = [None] * n
    work = [(0, frozenset())]
    hazards = {}
    while work:
        i, fl = work.pop()
        while i < n:
            if state_in[i] is not None and fl <= state_in[i]:
                break
            fl = fl | (state_in[i] or frozenset())
            state_in[i] = fl
            op, args, in_asm, text = ins[i]
            if in_asm and (op.startswith("s_load") or op.startswith("ds_read")):
                for a in args[1:]:
                    if regs(a) & fl:
                        hazards[i] = f"request READS in-flight registers: {text}"
                fl = fl | regs(args[0])
            elif op == "s_waitcnt" and "lgkmcnt(0)" in text:
                fl = frozenset()
            elif not op.startswith(NO_DEST) and args:
                dst = regs(args[0])
                if op.startswith("v_cmp") and len(args) > 2 and not op.endswith("_e32"):
                    pass
                if dst & fl:
                    hazards[i] = f"WRITES in-flight {sorted(dst & fl)[:4]}: {text}"
                for a in args[1:]:
                    if regs(a) & fl:
                        hazards[i] = f"READS in-flight {sorted(regs(a) & fl)[:4]}: {text}"
                        break
            elif op.startswith(("global_store", "ds_write", "v_cmpx", "s_cmp")):
                for a in args:
                    if regs(a) & fl:
                        hazards[i] = f"READS in-flight {sorted(regs(a) & fl)[:4]}: {text}"
                        break
            if op in ("s_branch", "s_setpc_b64"):
                if op == "s_branch" and args[0] in labels:
                    work.append((labels[args[0]], fl))
                break
            if op.startswith("s_cbranch") and args and args[-1] in labels:
                work.append((labels[args[-1]], fl))
            if op == "s_endpgm":
                break
            i += 1
    return [f"{name[:60]}: instr {i}: {msg}" for i, msg in sorted(hazards.items())]


def main():
    text = open(sys.argv[1]).read()
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    bad, seen = [], 0
    for name, lines in kernels(text):
        if want in name and any("ASMSTART" in l for l in lines):
            seen += 1
            bad += check(name, lines)
    for b in bad[:40]:
        print(b)
    print(f"{seen} kernel(s) with hand-issued requests checked, {len(bad)} hazard(s)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
