#!/usr/bin/env python3
"""Replays the hand-counted s_waitcnt vmcnt(N) of a kernel's hottest loop on its own disassembly: vector-memory loads complete in
issue order, so a wait for vmcnt <= N retires all but the N youngest.  Walks the loop body (twice: the state the back edge carries
in), keeps the queue of loads in flight with their destination registers, and reports every instruction that reads or writes a
register a load still in flight is going to write.  LDS-DMAs (`... lds`) have no register destination: for them the check is the
count at the s_barrier -- the number of loads issued after the youngest DMA that has to be visible -- printed for inspection.

    python tools/check_waits.py <file.s | library.so> <mangled or demangled kernel name substring>
"""
import re
import sys


def loop_body(lines):
    labels = {}
    for i, l in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    best = None
    for i, l in enumerate(lines):
        m = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            body = lines[labels[m.group(1)] : i + 1]
            n = sum("v_mfma" in x for x in body)
            if best is None or n > best[0]:
                best = (n, body)
    return best[1] if best else []


def tail_paths(lines, limit=64):
    """every control-flow path from the hottest loop's exit to the first full drain (s_waitcnt vmcnt(0)), as instruction lists:
    the ragged tail's guarded copies of a stage are laid out out of line, so the text order is not the execution order"""
    body = loop_body(lines)
    if not body:
        return [], []
    end = next(i for i in range(len(lines)) if lines[i : i + len(body)] == body) + len(body)
    labels = {m.group(1): i for i, l in enumerate(lines) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
    paths = []
    has_marker = any(l.split(";")[0].strip() == "s_nop 15" for l in lines[end:])

    def walk(i, acc, depth):
        while i < len(lines) and len(paths) < limit:
            t = lines[i].split(";")[0].strip()
            i += 1
            if not t or t.startswith(".") and not t.endswith(":") or t.endswith(":"):
                continue
            acc.append(t)
            # the end of the tail: the wide kernel marks it (`s_nop 15` in front of its epilogue's first accumulator read, behind the drain); without the
            # marker the first full drain.  (A drain INSIDE the tail -- hipcc's own wait for a scratch reload it placed there -- does not end the path.)
            if (t == "s_nop 15" if has_marker else re.fullmatch(r"s_waitcnt vmcnt\(0\)", t)) or t.startswith("s_endpgm"):
                paths.append(acc)
                return
            m = re.match(r"s_cbranch_\w+ (\.LBB\d+_\d+)", t)
            if m and depth < 12:
                walk(labels[m.group(1)], list(acc), depth + 1)
                continue
            m = re.match(r"s_branch (\.LBB\d+_\d+)", t)
            if m:
                i = labels[m.group(1)]
        paths.append(acc)

    walk(end, [], 0)

    def feasible(path):  # the tail's guards are monotone (stage u runs only if stage u - 1 ran): no MFMAs behind a skipped stage
        counts, n, seen = [], 0, False
        for t in path:
            if t.startswith("s_cmp_"):
                if seen:
                    counts.append(n)
                seen, n = True, 0
            elif t.startswith("v_mfma"):
                n += 1
        counts.append(n)
        return all(not (a == 0 and b > 0) for a, b in zip(counts, counts[1:]))

    return body, [p for p in paths if feasible(p)]


def regs(tok):
    """'v[12:15]' -> {12..15}; 'v7' -> {7}; anything else -> empty"""
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def check(body, tail=()):
    """body: the loop's lines; tail: one path behind it (instructions).  -> (problems, loads in flight at each barrier of the loop)"""
    def clean(ls):
        ins = [x.split(";")[0].strip() for x in ls]
        return [x for x in ins if x and not x.startswith(".") and not x.endswith(":")]
    insts, tail_insts = clean(body), clean(tail)
    queue = []  # [dest regs (set) or None for an LDS-DMA]
    lds = []    # LDS reads in flight (round 6: the fragment re-reads are inline asm too, their lgkmcnt waits hand-placed): destination registers, issue order = return order
    problems, barrier_counts = [], []
    for rep in range(3 if tail_insts else 2):
        for ins in (insts if rep < 2 else tail_insts):
            op, _, rest = ins.partition(" ")
            toks = [t.strip() for t in rest.split(",")]
            m = re.search(r"vmcnt\((\d+)\)", ins)
            ml = re.search(r"lgkmcnt\((\d+)\)", ins)
            if op == "s_waitcnt" and (m or ml):
                if m:
                    n = int(m.group(1))
                    if len(queue) > n:
                        queue = queue[len(queue) - n :]
                if ml:
                    n = int(ml.group(1))
                    lds = lds[len(lds) - n :] if n and len(lds) > n else ([] if n == 0 else lds)
                continue
            used = set()
            for t in toks:
                used |= regs(t.split()[0] if t else "")
            for q in lds:  # reads or overwrites the destination of an LDS read that has not returned
                if q & used:
                    problems.append("LDS: " + ins)
            if op.startswith("ds_read"):
                lds.append(regs(toks[0]))
                continue
            if op.startswith("buffer_load") or op.startswith("global_load"):
                if " lds" in ins:
                    queue.append(None)
                    continue
                dest = regs(toks[0])
                addr = set()
                for t in toks[1:]:
                    addr |= regs(t.split()[0] if t else "")
                for q in queue:
                    if q and (q & (dest | addr)):
                        problems.append(ins)
                queue.append(dest)
                continue
            if op == "s_barrier" and rep == 1:
                # loads in flight at the barrier, oldest first: D = LDS-DMA, r = register load
                barrier_counts.append("".join("D" if q is None else "r" for q in queue))
            for q in queue:
                if q and (q & used):
                    problems.append(ins)
    return problems, barrier_counts


def check_prologue(lines):
    """the asm loads (buffer_load ...) in front of the loop: nothing may touch their destinations before the wait that drains
    them.  (The compiler's own global_loads up there are its business: it counts them itself.)"""
    body = loop_body(lines)
    if not body:
        return []
    start = next(i for i in range(len(lines)) if lines[i : i + len(body)] == body)
    queue, problems = [], []
    for l in lines[:start]:
        ins = l.split(";")[0].strip()
        if not ins or ins.startswith(".") or ins.endswith(":"):
            continue
        op, _, rest = ins.partition(" ")
        toks = [t.strip() for t in rest.split(",")]
        m = re.search(r"vmcnt\((\d+)\)", ins)
        if op == "s_waitcnt" and m:
            n = int(m.group(1))
            queue = queue[len(queue) - n :] if n and len(queue) > n else ([] if n == 0 else queue)
            continue
        if op.startswith("buffer_load"):
            queue.append(None if " lds" in ins else regs(toks[0]))
            continue
        used = set()
        for t in toks:
            used |= regs(t.split()[0] if t else "")
        problems += [ins for q in queue if q and (q & used)]
    return problems


def main():
    src, name = sys.argv[1], sys.argv[2]
    if src.endswith(".so"):
        import os
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import code_object as co
        sym = [k["name"] for k in co.kernels(src) if name in (k["name"], k["demangled"])]
        text = co.disassemble(src, sym[0])
    else:
        text = open(src).read()
    body, paths = tail_paths(text.split("\n"))
    problems, barrier = check(body)
    for path in paths:
        problems += check(body, path)[0]
    problems = sorted(set(problems + check_prologue(text.split("\n"))))
    print(f"{len(paths)} paths from the loop exit to the drain")
    print(f"{sum('v_mfma' in x for x in body)} MFMAs in the loop; in flight at each s_barrier (oldest first): {barrier}")
    for p in problems:
        print("TOUCHES A LOAD IN FLIGHT:", p)
    return 1 if problems else 0


if __name__ == "__main__":
    sys.exit(main())
