"""Per-kernel resource table of a built library, read from the gfx950 code object's metadata (no GPU needed).

    python tools/code_object.py [qqq_amd/libqqq_amd.so]

Columns: VGPRs (arch + acc), spilled VGPRs, scratch bytes per lane, LDS bytes (static), SGPRs.  tests/test_code_object_cpu.py
holds the product library to "no scratch in any kernel the dispatcher can pick on its own"."""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
FIELDS = ("vgpr_count", "agpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
          "group_segment_fixed_size", "sgpr_count", "max_flat_workgroup_size")


def _demangle(sym):
    """`_Z16qqq_panel_kernelILi8ELb0E...E<args>` -> `qqq_panel_kernel<8,false,...>` (integral / bool template arguments only)."""
    m = re.match(r"_Z(\d+)", sym)
    if not m:
        return sym
    n = int(m.group(1))
    name, rest = sym[m.end():m.end() + n], sym[m.end() + n:]
    if not rest.startswith("I"):
        return name
    args, i = [], 1
    while rest[i] == "L":
        a = re.match(r"L([ib])(n?\d+)E", rest[i:])
        v = a.group(2).replace("n", "-")
        args.append(("true" if v != "0" else "false") if a.group(1) == "b" else v)
        i += a.end()
    return f"{name}<{','.join(args)}>"


def kernels(lib):
    """[{name, demangled, vgpr_count, ...}] for every kernel in the gfx950 bundle of `lib`."""
    lib = os.path.abspath(lib)
    with tempfile.TemporaryDirectory() as td:
        link = os.path.join(td, "lib.so")
        os.symlink(lib, link)
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", link], cwd=td, check=True, capture_output=True)
        cos = [f for f in os.listdir(td) if "gfx950" in f]
        if not cos:
            raise RuntimeError(f"no gfx950 code object in {lib}")
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", os.path.join(td, cos[0])], check=True, capture_output=True, text=True).stdout
    out = []
    # one "- .agpr_count:" ... block per kernel inside amdhsa.kernels
    body = notes.split("amdhsa.kernels:", 1)[1].split("amdhsa.target:", 1)[0]
    for blk in re.split(r"\n\s+- \.agpr_count:", "\n" + body)[1:]:
        blk = ".agpr_count:" + blk
        d = {}
        m = re.search(r"^\s*\.name:\s+(\S+)", blk, re.M)
        if not m:
            continue
        d["name"] = m.group(1).strip("'\"")
        for f in FIELDS:
            m = re.search(rf"\.{f}:\s+(\d+)", blk)
            d[f] = int(m.group(1)) if m else 0
        out.append(d)
    for k in out:
        k["demangled"] = _demangle(k["name"])
    return out


def disassemble(lib, symbol):
    """Disassembly of kernel `symbol` (mangled) as compiler-style text: one instruction per line, `.LBB0_<n>:` labels at the
    branch targets and the branches rewritten to name them (what tools/check_waits.py walks)."""
    lib = os.path.abspath(lib)
    with tempfile.TemporaryDirectory() as td:
        link = os.path.join(td, "lib.so")
        os.symlink(lib, link)
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", link], cwd=td, check=True, capture_output=True)
        co = [f for f in os.listdir(td) if "gfx950" in f][0]
        dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", f"--disassemble-symbols={symbol}", os.path.join(td, co)],
                             check=True, capture_output=True, text=True).stdout
    ins = []
    for line in dis.split("\n"):
        m = re.match(r"\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):(.*)", line)
        if m:
            ins.append((int(m.group(3), 16), m.group(1), m.group(2), m.group(4)))
    if not ins:
        raise RuntimeError(f"{symbol} not found in {lib}")
    base = ins[0][0]
    targets = {}
    for addr, op, args, tail in ins:
        m = re.search(r"\+0x([0-9a-fA-F]+)>", tail) if op.startswith(("s_cbranch", "s_branch")) else None
        if m:
            targets.setdefault(base + int(m.group(1), 16), len(targets))
    out = []
    for addr, op, args, tail in ins:
        if addr in targets:
            out.append(f".LBB0_{targets[addr]}:")
        m = re.search(r"\+0x([0-9a-fA-F]+)>", tail) if op.startswith(("s_cbranch", "s_branch")) else None
        out.append(f"\t{op} .LBB0_{targets[base + int(m.group(1), 16)]}" if m else f"\t{op} {args}")
    return "\n".join(out)


def hottest_loop(lib, symbol):
    """Instruction mix of the loop with the most MFMAs in kernel `symbol` (mangled name) of `lib`: {opcode: count} over the
    instructions between a backward branch and its target, plus "waits" = the s_waitcnt operands seen inside.  The loop is
    found from the disassembly alone (a backward s_cbranch), so this needs neither a GPU nor compiler remarks."""
    lib = os.path.abspath(lib)
    with tempfile.TemporaryDirectory() as td:
        link = os.path.join(td, "lib.so")
        os.symlink(lib, link)
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", link], cwd=td, check=True, capture_output=True)
        co = [f for f in os.listdir(td) if "gfx950" in f][0]
        dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", f"--disassemble-symbols={symbol}", os.path.join(td, co)],
                             check=True, capture_output=True, text=True).stdout
    ins = []  # (address, opcode, operands)
    for line in dis.split("\n"):
        m = re.match(r"\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):(.*)", line)
        if m:
            ins.append((int(m.group(3), 16), m.group(1), m.group(2), m.group(4)))
    if not ins:
        raise RuntimeError(f"{symbol} not found in {lib}")
    base = ins[0][0]
    best = None
    for addr, op, args, tail in ins:
        if not op.startswith("s_cbranch"):
            continue
        m = re.search(r"\+0x([0-9a-fA-F]+)>", tail)
        if not m:
            continue
        tgt = base + int(m.group(1), 16)
        if tgt >= addr:
            continue  # forward
        body = [x for x in ins if tgt <= x[0] <= addr]
        n = sum(1 for x in body if x[1].startswith("v_mfma"))
        if best is None or n > best[0]:
            best = (n, body)
    mix = {}
    waits = []
    for _, op, args, _t in (best[1] if best else []):
        mix[op] = mix.get(op, 0) + 1
        if op == "s_waitcnt":
            waits.append(args)
    mix["waits"] = waits
    return mix


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "qqq_amd", "libqqq_amd.so")
    ks = kernels(lib)
    print(f"{len(ks)} kernels in {lib}")
    print(f"{'kernel':60s} {'vgpr':>5s} {'agpr':>5s} {'spill':>5s} {'scratch':>7s} {'lds':>7s} {'sgpr':>5s}")
    for k in sorted(ks, key=lambda k: k["demangled"]):
        print(f"{k['demangled'][:60]:60s} {k['vgpr_count']:5d} {k['agpr_count']:5d} {k['vgpr_spill_count']:5d} "
              f"{k['private_segment_fixed_size']:7d} {k['group_segment_fixed_size']:7d} {k['sgpr_count']:5d}")
