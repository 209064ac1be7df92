#!/usr/bin/env python3
"""Whole-kernel replay of the vector-memory wait counts of a compiled gfx950 kernel, over its control-flow graph (CPU only).

tools/check_waits.py walks ONE loop body and the straight paths behind it; the persistent tile walk of the wide kernel (round 4)
has a seam behind every stage of its trip and no tail, so the check has to follow branches.  This one builds the CFG of the
disassembly (labels, s_cbranch_*, s_branch, s_endpgm) and pushes an abstract state through it to a fixpoint:

    state = the queue of vector-memory LOADS in flight, oldest first; each entry is the set of registers the load will write
            (empty for an LDS-DMA `... lds`), capped at the 63 a 6-bit vmcnt can hold (the hardware stalls the issue instead).

Loads retire in issue order, so `s_waitcnt vmcnt(N)` keeps the N youngest entries.  Every instruction that reads or writes a
register a load still in flight is going to write is reported -- a hand-counted wait that is too lax, a compiler-inserted copy
or spill of a ring register whose load has not landed, the destination of a dead asm load handed to another value.

STORES are not queued.  On gfx9 they do occupy vmcnt slots (in order with the loads), so leaving them out makes every later
wait retire FEWER loads than the hardware would: the replay is conservative (it can flag code whose waits rely on counting a
store, never miss a hazard), and the guarded stores of an epilogue do not multiply the states.  `--count-stores` queues them
(exact in-order model, exponential in the number of exec-guarded store blocks: for small kernels only).

Also reported: writers of M0 other than the wide kernel's own (`s_add_u32 m0` / `s_mov_b32 m0` inside its asm statements are
listed so that a test can assert nothing else touches it between the M0 write and the LDS-DMA that reads it).

    python tools/check_vmem.py <file.s | library.so> <kernel name substring> [--count-stores]
"""
import re
import sys

VMEM_LOAD = ("buffer_load", "global_load", "scratch_load", "flat_load")
VMEM_STORE = ("buffer_store", "global_store", "scratch_store", "flat_store", "buffer_atomic", "global_atomic", "flat_atomic")
CAP = 63


def parse(text):
    """-> [(labels (tuple), op, operand string)] for one kernel's instructions (compiler .s or code_object.disassemble output)"""
    out = []
    pending = ()
    for raw in text.split("\n"):
        line = raw.split(";")[0].split("//")[0].rstrip()
        if not line.strip():
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", line.strip())
        if m:
            pending = pending + (m.group(1),)
            continue
        t = line.strip()
        if t.startswith(".") or t.endswith(":"):
            continue
        op, _, rest = t.partition(" ")
        out.append((pending, op, rest.strip()))
        pending = ()
    return out


def kernel_text(src, name):
    if src.endswith(".so"):
        import os
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import code_object as co
        syms = [k["name"] for k in co.kernels(src) if name in (k["name"], k["demangled"])]
        if not syms:
            raise SystemExit(f"no kernel {name} in {src}")
        return co.disassemble(src, syms[0])
    text = open(src).read()
    # a compiler .s may hold several kernels: cut from the kernel's label to its s_endpgm-terminated function end (.Lfunc_end)
    m = re.search(r"^(\S*%s\S*):\s*(?:;.*)?$" % re.escape(name), text, re.M)
    if not m:
        return text
    start = m.end()
    e = re.search(r"^\.Lfunc_end\d+:", text[start:], re.M)
    return text[start:start + e.start()] if e else text[start:]


def regs(tok):
    """'v[12:15]' -> {('v',12..15)}; 'a7' -> {('a',7)}; anything else -> empty"""
    tok = tok.strip()
    m = re.fullmatch(r"([va])\[(\d+):(\d+)\]", tok)
    if m:
        return {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.fullmatch(r"([va])(\d+)", tok)
    return {(m.group(1), int(m.group(2)))} if m else set()


def operand_regs(rest):
    used = set()
    for t in rest.split(","):
        t = t.strip()
        if t:
            used |= regs(t.split()[0])
    return used


def build_blocks(ins):
    """basic blocks: [(first, last+1)], successor lists by block index"""
    label_at = {l: i for i, (lab, _, _) in enumerate(ins) for l in lab}
    leaders = {0}
    for i, (lab, op, rest) in enumerate(ins):
        if lab:
            leaders.add(i)
        if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")) and i + 1 < len(ins):
            leaders.add(i + 1)
    starts = sorted(leaders)
    blk_of = {}
    blocks = []
    for bi, s in enumerate(starts):
        e = starts[bi + 1] if bi + 1 < len(starts) else len(ins)
        blocks.append((s, e))
        blk_of[s] = bi
    succ = []
    for (s, e) in blocks:
        lab, op, rest = ins[e - 1]
        nxt = []
        if op.startswith("s_endpgm") or op.startswith("s_setpc"):
            pass
        elif op.startswith("s_branch"):
            nxt = [blk_of[label_at[rest.split()[0]]]]
        elif op.startswith("s_cbranch"):
            nxt = [blk_of[label_at[rest.split()[0]]]]
            if e < len(ins):
                nxt.append(blk_of[e])
        elif e < len(ins):
            nxt = [blk_of[e]]
        succ.append(nxt)
    return blocks, succ


def sregs(tok):
    """'s[24:25]' -> {24, 25}; 's7' -> {7}; 'vcc' -> {'vcc'}; else empty"""
    tok = tok.strip()
    m = re.fullmatch(r"s\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"s(\d+)", tok)
    if m:
        return {int(m.group(1))}
    return {"vcc"} if tok in ("vcc", "vcc_lo", "vcc_hi") else set()


def step_flags(flags, op, rest):
    """flags: {register pair text or 'vcc': 0 | -1} -- the exit / guard flags the structurizer keeps in SGPR pairs
    (`s_mov_b64 s[24:25], -1` ... `s_and_b64 vcc, exec, s[24:25]` ... `s_cbranch_vccnz`): enough constant propagation to drop
    the paths those flags rule out (a `return` inside the loop is compiled into such a flag and a jump through the latch)."""
    toks = [t.strip() for t in rest.split(",")] if rest else []
    if op == "s_mov_b64" and len(toks) == 2 and toks[1] in ("0", "-1") and re.fullmatch(r"s\[\d+:\d+\]", toks[0]):
        flags = {k: v for k, v in flags.items() if k == "vcc" or not (sregs(k) & sregs(toks[0]))}
        flags[toks[0]] = int(toks[1])
        return flags
    if op in ("s_and_b64", "s_andn2_b64") and len(toks) == 3 and toks[0] == "vcc" and toks[1] == "exec" and toks[2] in flags:
        flags = dict(flags)
        v = flags[toks[2]]
        flags["vcc"] = v if op == "s_and_b64" else (-1 - v)  # (exec is not zero where a wave executes)
        return flags
    if op.startswith(("s_cbranch", "s_branch", "s_waitcnt", "s_nop", "s_barrier", "s_endpgm", "s_cmp", "s_bitcmp", "s_sleep", "s_setprio")):
        return flags
    if not flags:
        return flags
    written = set()
    for t in toks[:2]:  # destination (and the carry-out / second destination of the _co_ forms)
        written |= sregs(t.split()[0] if t else "")
    if "vcc" in rest:
        written.add("vcc")
    if op.startswith("v_") and not written:
        return flags
    return {k: v for k, v in flags.items() if not ((sregs(k) if k != "vcc" else {"vcc"}) & written)}


def run(ins, count_stores=False, max_states=200000):
    blocks, succ = build_blocks(ins)
    label_at = {l: i for i, (lab, _, _) in enumerate(ins) for l in lab}
    blk_at = {s: bi for bi, (s, e) in enumerate(blocks)}
    problems = {}
    m0_writers = []
    seen = [set() for _ in blocks]
    work = [(0, (), ())]
    nstates = 0
    max_queue = 0
    while work:
        bi, state, fl = work.pop()
        if (state, fl) in seen[bi]:
            continue
        seen[bi].add((state, fl))
        nstates += 1
        if nstates > max_states:
            raise SystemExit(f"more than {max_states} (block, queue) states: the replay does not converge")
        queue = list(state)
        flags = dict(fl)
        s, e = blocks[bi]
        for i in range(s, e):
            lab, op, rest = ins[i]
            flags = step_flags(flags, op, rest)
            if op == "s_waitcnt":
                m = re.search(r"vmcnt\((\d+)\)", rest)
                if m:
                    n = int(m.group(1))
                    if len(queue) > n:
                        queue = queue[len(queue) - n:] if n else []
                continue
            if op.startswith("s_waitcnt"):
                continue
            used = operand_regs(rest)
            if op.startswith("v_pk_") and "op_sel_hi:[0,1]" in rest and "op_sel:" not in rest:
                # a packed op that BROADCASTS the low half of src0 (op_sel_hi:[0,1]: both results take src0.lo): the encoding names the aligned pair v[n:n+1], the
                # value of v[n+1] is not used -- hipcc parks a scalar factor next to whatever register is free, e.g. a ring word whose load is still in flight (round 6, dword loads)
                toks_ = [t.strip() for t in rest.split(",")]
                if len(toks_) >= 2:
                    pair = regs(toks_[1].split()[0])
                    if len(pair) == 2:
                        others = set()
                        for j, t_ in enumerate(toks_):
                            if j != 1 and t_:
                                others |= regs(t_.split()[0])
                        used = others | {min(pair)}
            if used:
                for q in queue:
                    if q and (q & used):
                        problems.setdefault(i, f"{op} {rest}")
                        break
            if op.startswith(VMEM_LOAD):
                toks = [t.strip() for t in rest.split(",")]
                dest = frozenset() if re.search(r"\blds\b", rest) else frozenset(regs(toks[0].split()[0]))
                queue.append(dest)
                if len(queue) > CAP:
                    queue = queue[len(queue) - CAP:]
            elif count_stores and op.startswith(VMEM_STORE):
                queue.append(frozenset())
                if len(queue) > CAP:
                    queue = queue[len(queue) - CAP:]
            max_queue = max(max_queue, len(queue))
        out = tuple(queue)
        nxt = succ[bi]
        lab, op, rest = ins[e - 1]
        if op in ("s_cbranch_vccz", "s_cbranch_vccnz") and "vcc" in flags:  # a branch on a known flag: one successor
            taken = (flags["vcc"] == 0) == (op == "s_cbranch_vccz")
            nxt = [blk_at[label_at[rest.split()[0]]]] if taken else ([blk_at[e]] if e < len(ins) else [])
        fo = tuple(sorted(flags.items()))
        for nb in nxt:
            if (out, fo) not in seen[nb]:
                work.append((nb, out, fo))
    for i, (lab, op, rest) in enumerate(ins):
        toks = [t.strip() for t in rest.split(",")]
        if toks and toks[0] == "m0" and op.startswith("s_"):
            m0_writers.append((i, f"{op} {rest}"))
    return problems, dict(states=nstates, blocks=len(blocks), max_queue=max_queue, m0_writers=m0_writers)


def scalar_peek(ins, i):
    """`v_readfirstlane_b32 sX, vY` of a register with a load in flight.  hipcc materialises the undefined inputs of the exit
    path's phis this way (the structurizer turns a loop exit into flags and phis; an undefined scalar is read from whatever
    VGPR is at hand).  It only READS the register: the load is not disturbed, and nothing in these kernels takes a scalar out
    of loaded data -- the hazards this tool exists for (a consumer running ahead of its load; a copy, spill or re-use of a
    register whose load has not landed) all show up as vector instructions.  Listed, not counted."""
    return ins[i][1] == "v_readfirstlane_b32"


def m0_discipline(ins, info):
    """every LDS-DMA must read an M0 written by the instruction group right in front of it: between an M0 write and the next
    `... lds` load no OTHER M0 write, and no LDS-DMA without a preceding M0 write in the same block or its dominating slot.
    (Linear scan in text order inside basic blocks: the wide kernel writes M0 one issue slot ahead of the DMA, in the same block.)"""
    bad = []
    last_m0 = None
    for i, (lab, op, rest) in enumerate(ins):
        if lab:
            last_m0 = None
        toks = [t.strip() for t in rest.split(",")]
        if toks and toks[0] == "m0" and op.startswith("s_"):
            last_m0 = i
        elif op.startswith(VMEM_LOAD) and re.search(r"\blds\b", rest):
            if last_m0 is None:
                bad.append((i, f"LDS-DMA without an M0 write in its block: {op} {rest}"))
            last_m0 = None
        elif op.startswith(("s_cbranch", "s_branch")):
            last_m0 = None
    return bad


def sgpr_vmem_hazards(ins, need=5):
    """A VALU instruction that writes an SGPR (v_readfirstlane / v_readlane / v_cmp ... with an SGPR destination) needs 5 wait
    states before a vector-memory instruction reads that SGPR (descriptor, scalar offset).  hipcc pads this itself, but counts
    an inline-asm statement between the two for more than it is worth: `v_readfirstlane s0, v4 / <asm MFMA> / s_nop 0 /
    buffer_load ... s0` reads the OLD s0 on the device (round 4: stale offsets in the first load behind every such copy).
    Cycles counted here: s_nop N = N + 1, a VALU instruction other than an MFMA = 4 (wave64 on a 16-lane SIMD), anything else
    = 1 -- the one measured data point either way (MFMA + s_nop 0 fails, MFMA + v_perm works)."""
    bad = []
    for i, (lab, op, rest) in enumerate(ins):
        if not op.startswith(VMEM_LOAD + VMEM_STORE):
            continue
        need_regs = set()
        for t in rest.split(","):
            need_regs |= {r for r in sregs(t.strip().split()[0] if t.strip() else "") if r != "vcc"}
        cycles, j = 0, i - 1
        while j >= 0 and cycles < need and need_regs:
            l2, op2, rest2 = ins[j]
            if op2.startswith(("s_cbranch", "s_branch", "s_barrier", "s_endpgm")):
                break
            toks = [t.strip() for t in rest2.split(",")] if rest2 else []
            if op2.startswith("v_") and toks:
                w = {r for r in sregs(toks[0].split()[0]) if r != "vcc"}
                if ("_co_" in op2 or op2.startswith(("v_div_scale", "v_mad_u64", "v_mad_i64"))) and len(toks) > 1:
                    w |= {r for r in sregs(toks[1].split()[0]) if r != "vcc"}
                if w & need_regs:
                    bad.append((i, f"{op} {rest}   <- {op2} {rest2}: {cycles} wait states"))
                    break
            m = re.fullmatch(r"(\d+)", rest2.strip()) if op2 == "s_nop" else None
            cycles += int(m.group(1)) + 1 if m else (4 if op2.startswith("v_") and not op2.startswith("v_mfma") else 1)
            if ins[j][0]:
                break  # a label: other paths join here, stop (conservative enough for straight-line slots)
            j -= 1
    return bad


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    src, name = args[0], args[1]
    ins = parse(kernel_text(src, name))
    problems, info = run(ins, count_stores="--count-stores" in sys.argv)
    peeks = {i for i in problems if scalar_peek(ins, i)}
    for i in sorted(peeks):
        print(f"note: scalar read of a register with a load in flight (undefined phi input): [{i}] {problems[i]}")
    problems = {i: t for i, t in problems.items() if i not in peeks}
    print(f"{len(ins)} instructions, {info['blocks']} blocks, {info['states']} (block, queue) states, at most {info['max_queue']} loads in flight")
    print(f"{len(info['m0_writers'])} writers of M0")
    for i, t in sorted(problems.items()):
        print(f"TOUCHES A LOAD IN FLIGHT: [{i}] {t}")
    bad = m0_discipline(ins, info)
    for i, t in bad:
        print(f"M0: [{i}] {t}")
    hz = sgpr_vmem_hazards(ins)
    for i, t in hz:
        print(f"VALU-SGPR -> VMEM: [{i}] {t}")
    return 1 if problems or bad or hz else 0


if __name__ == "__main__":
    sys.exit(main())
