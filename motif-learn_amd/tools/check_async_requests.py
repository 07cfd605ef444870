#!/usr/bin/env python3
"""Static check of kernels that request operands through volatile asm (zk_sep_strip.hip: s_load / ds_read issued by hand,
waited for by hand).  The compiler takes an asm's outputs for written at once; the hardware writes them LATER.  Until the
s_waitcnt lgkmcnt(0) that covers a request, nothing else may write or read its destination registers -- in particular no
rematerialised s_load of a kernel argument into a register set the compiler believes dead (that race, a pointer overwritten by
late table data, was a memory-aperture fault in round 3).

  check_async_requests.py file.s [kernel-name-substring]      exit 1 and a listing if any hazard exists

Walks the control-flow graph of every kernel in the assembly (hipcc -S --cuda-device-only): the QUEUE of operations that may
still be outstanding on the LGKM counter is propagated along fall-through and branch edges; s_waitcnt lgkmcnt(N) with N > 0
retires what the in-order return of LDS reads guarantees (zk_frame_strip3_kernel waits for a pixel pair with the next pair
still in flight), lgkmcnt(0) everything."""
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
    # State = the queue of LGKM operations that may still be outstanding, oldest first: (kind, destination registers, hand-issued).
    # LDS reads return in order, scalar loads in any order, and s_waitcnt lgkmcnt(N) returns when at most N are outstanding:
    #   an LDS entry is certainly complete iff more than N LDS entries stand at or behind it in the queue (were it
    #   outstanding, all of those would be too); a scalar entry only after lgkmcnt(0).
    # The compiler's own loads count (they share the counter) but only hand-issued destinations are checked: the compiler
    # places waits for what it issued itself.  Every distinct state reaching an instruction is followed (bounded).
    n = len(ins)
    seen_states = [set() for _ in range(n)]
    work = [(0, ())]
    hazards = {}

    def in_flight(q):
        out = set()
        for kind, rg, hand in q:
            if hand:
                out |= rg
        return out

    while work:
        i, q = work.pop()
        while i < n:
            if q in seen_states[i]:
                break
            if len(seen_states[i]) > 256:
                hazards[i] = "too many distinct request states reach this instruction (checker limit)"
                break
            seen_states[i].add(q)
            fl = in_flight(q)
            op, args, in_asm, text = ins[i]
            is_lds_read = op.startswith("ds_read")
            is_smem = op.startswith("s_load") or op.startswith("s_buffer_load") or op.startswith("s_memtime") or op.startswith("s_memrealtime")
            if is_lds_read or is_smem:
                for a in args[1:]:
                    if regs(a) & fl:
                        hazards[i] = f"request READS in-flight registers: {text}"
                if regs(args[0]) & fl:
                    hazards[i] = f"request WRITES in-flight registers: {text}"
                q = q + (("lds" if is_lds_read else "smem", frozenset(regs(args[0])), bool(in_asm)),)
            elif op.startswith(("ds_write", "ds_add", "ds_swizzle", "ds_bpermute", "ds_permute", "s_sendmsg")) or op.startswith("ds_"):
                for a in args:
                    if regs(a) & fl:
                        hazards[i] = f"READS in-flight {sorted(regs(a) & fl)[:4]}: {text}"
                        break
                q = q + (("lds", frozenset(), False),)   # occupies the counter, returns in order with the reads
            elif op == "s_waitcnt":
                m = re.search(r"lgkmcnt\((\d+)\)", text)
                if m:
                    N = int(m.group(1))
                    lds_behind = 0
                    keep = []
                    for kind, rg, hand in reversed(q):
                        if kind == "lds":
                            lds_behind += 1
                            if lds_behind <= N:
                                keep.append((kind, rg, hand))
                        elif N > 0:
                            keep.append((kind, rg, hand))
                    q = tuple(reversed(keep))
            elif not op.startswith(NO_DEST) and args:
                dst = regs(args[0])
                if dst & fl:
                    hazards[i] = f"WRITES in-flight {sorted(dst & fl)[:4]}: {text}"
                for a in args[1:]:
                    if regs(a) & fl:
                        hazards[i] = f"READS in-flight {sorted(regs(a) & fl)[:4]}: {text}"
                        break
            elif op.startswith(("global_store", "v_cmpx", "s_cmp")):
                for a in args:
                    if regs(a) & fl:
                        hazards[i] = f"READS in-flight {sorted(regs(a) & fl)[:4]}: {text}"
                        break
            if op in ("s_branch", "s_setpc_b64"):
                if op == "s_branch" and args[0] in labels:
                    work.append((labels[args[0]], q))
                break
            if op.startswith("s_cbranch") and args and args[-1] in labels:
                work.append((labels[args[-1]], q))
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
