#!/usr/bin/env python3
"""(CPU) Issue-slot load of the wide kernel's compiled steady-state loop: the instructions between consecutive MFMAs ("extras" of a slot), their histogram, and
the cycles per MFMA the lone-wave issue model of tools/mfma_issue_bench.hip predicts (an instruction ~5.15 cycles of issue, a 16.3-cycle MFMA hides two; a slot with n
extras costs max(16.3, 5.15 (n + 1)); ds_read_b128 / LDS-DMA counted as ONE instruction here, i.e. a lower bound).
    python tools/slot_load.py [lib.so] [kernel ...]"""
import collections, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import code_object

DEFAULT = ("qqq_wide_kernel<0,16,4,8,2,false>", "qqq_wide_kernel<1,16,4,4,2,false>", "qqq_wide_kernel<2,16,4,4,2,false>", "qqq_wide_kernel<0,8,4,8,2,false>",
           "qqq_wide_kernel<0,16,4,8,1,false>", "qqq_wide_kernel<0,16,4,4,2,true>")


def hottest_loop_lines(lib, sym):
    d = code_object.disassemble(lib, sym).split("\n")
    labels = {l.rstrip(":"): i for i, l in enumerate(d) if l.startswith(".LBB")}
    best = None
    for i, l in enumerate(d):
        m = re.match(r"\ts_cbranch\S+ (\.LBB\S+)", l)
        if m and labels.get(m.group(1), 1 << 30) < i:
            body = d[labels[m.group(1)]:i + 1]
            n = sum("v_mfma" in x for x in body)
            if best is None or n > best[0]:
                best = (n, body)
    return best


def slots(body):
    out, cur = [], []
    for l in body:
        if l.startswith(".LBB"):
            continue
        op = l.split()[0]
        if op.startswith("v_mfma"):
            out.append(cur)
            cur = []
        else:
            cur.append(op)
    out.append(cur)
    return out[1:]


def report(lib, name, sym, c=5.15, mf=16.3):
    n, body = hottest_loop_lines(lib, sym)
    sl = slots(body)
    hist = collections.Counter(len(s) for s in sl)
    loss = sum(max(0.0, (1 + len(s)) * c - mf) for s in sl)
    extras = sum(len(s) for s in sl)
    print(f"{name}: {n} MFMAs, {extras} other instructions ({extras / n:.2f} per MFMA); slots by number of extras {sorted(hist.items())}; model {mf + loss / n:.2f} cycles per MFMA")
    return sl


if __name__ == "__main__":
    args = sys.argv[1:]
    lib = args.pop(0) if args and args[0].endswith(".so") else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "qqq_amd", "libqqq_amd.so")
    ks = {k["demangled"]: k["name"] for k in code_object.kernels(lib)}
    for nm in (args or DEFAULT):
        if nm in ks:
            sl = report(lib, nm, ks[nm])
            if os.environ.get("HEAVY"):
                for s in sl:
                    if len(s) >= int(os.environ["HEAVY"]):
                        print("   ", s)
