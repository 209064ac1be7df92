"""Static resource check of the built gfx950 code object (no GPU): which kernels carry scratch, and how many registers
the hot instantiations take.  A register spill inside a main loop is the difference between the measured numbers in
profiles/ and several times slower (e.g. the 4-stage activation ring of the 64-column panel shape: 304 B of scratch,
4.5x), so a compiler or source change that introduces one should fail here, before it reaches the GPU box."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

# Instantiations known to keep a few registers in scratch, all OUTSIDE their main loop (checked in the ISA: the
# scratch_store sits in front of the loop, the scratch_load behind it) or forced test variants the dispatcher never
# picks on its own (the 8-deep prefetch rings of the small-m panel shapes, the 4-stage per-group 64-column ring).
KNOWN_SCRATCH = {
    "qqq_panel_kernel<1,false,8,1,8,8,1>", "qqq_panel_kernel<1,true,8,1,8,8,1>",
    "qqq_panel_kernel<2,false,8,1,8,8,1>", "qqq_panel_kernel<2,true,8,1,8,8,1>",
    "qqq_panel_kernel<8,false,4,2,3,3,2>", "qqq_panel_kernel<8,true,4,2,3,3,2>", "qqq_panel_kernel<8,true,4,2,4,2,2>",
    "qqq_panel_kernel<8,true,4,2,4,4,2>",
    "qqq_panel_kernel<8,true,8,1,4,4,1>", "qqq_tiled_kernel<256,8,1,1,true,0>",
}
# The wide kernel (one wave per SIMD, 256 + 256 registers) parks values that are live across its main loop but not used
# in it in scratch: stores in front of the loop, loads behind it -- test_steady_state_loops holds the loop itself to zero.
WIDE_SCRATCH_BYTES = 128


@pytest.fixture(scope="module")
def table():
    import code_object
    from qqq_amd import build

    build.build()
    return {k["demangled"]: k for k in code_object.kernels(build.LIB)}


def test_every_family_is_in_the_code_object(table):
    fams = {n.split("<")[0] for n in table}
    assert {"qqq_column_kernel", "qqq_stream_kernel", "qqq_panel_kernel", "qqq_tiled_kernel", "qqq_reduce_kernel",
            "qqq_dynamic_quant_kernel", "qqq_pack_int4_kernel", "qqq_unpack_int4_kernel"} <= fams


def test_scratch_only_where_known(table):
    spill = {n for n, k in table.items() if k["private_segment_fixed_size"] or k["vgpr_spill_count"] or k["sgpr_spill_count"]}
    wide = {n for n in spill if n.startswith("qqq_wide_kernel")}
    assert spill - wide <= KNOWN_SCRATCH, sorted(spill - wide - KNOWN_SCRATCH)
    for n in spill:
        assert table[n]["private_segment_fixed_size"] <= (WIDE_SCRATCH_BYTES if n in wide else 128), (n, table[n])


def test_hot_instantiations(table):
    # the kernels behind the BASELINE sweep's bench points: per-channel decode column kernel, the M=16 stream kernel,
    # the M=128 panel shape, the 64-column panel shape (M >= 768) and the per-group large-m tiled tile
    x2 = table["qqq_panel_kernel<8,false,4,2,4,2,2>"]
    assert x2["private_segment_fixed_size"] == 0 and x2["vgpr_count"] <= 256 and x2["agpr_count"] == 0  # 2 waves / SIMD
    m128 = table["qqq_panel_kernel<8,false,4,2,4,4,1>"]
    assert m128["private_segment_fixed_size"] == 0 and m128["vgpr_count"] <= 256
    # the 256 x 256 tiles of the M = 4096 points: the ring depth the dispatcher takes per mode is the instantiation WITHOUT the epilogue
    # spills (one VGPR parked across the loop is all that is left; the other depth measured 1.4-1.7 % slower: profiles/r05_wide_ring_depth.txt)
    for n in ("qqq_wide_kernel<0,16,4,8,2,false>", "qqq_wide_kernel<1,16,4,4,2,false>"):
        assert table[n]["vgpr_spill_count"] <= 12 and table[n]["private_segment_fixed_size"] <= 48, (n, table[n])  # (round 6, exchange hand-off in the epilogue: 8 - 10 registers, both ring depths alike)
    for n, k in table.items():
        if n.startswith(("qqq_column_kernel", "qqq_stream_kernel", "qqq_dynamic_quant_kernel", "qqq_reduce_kernel")):
            assert k["private_segment_fixed_size"] == 0, n
        if n.startswith("qqq_tiled_kernel<256,") and ",false," in n:
            assert k["private_segment_fixed_size"] == 0 and k["vgpr_count"] <= 256, n
        # one workgroup must fit a CU: 512 registers per lane and SIMD, 8-wave workgroups -> 2 waves per SIMD
        assert k["vgpr_count"] <= 512, n
        if n.startswith("qqq_wide_kernel"):  # 4-wave workgroups, one wave per SIMD: all 256 accumulation registers
            args = n.split("<")[1].rstrip(">").split(",")
            mt, hw = int(args[1]), int(args[4])  # 16 / 8 m-tiles x 2 hw column sets x 4 registers
            assert k["agpr_count"] == 8 * hw * mt and k["max_flat_workgroup_size"] == 256, (n, k)
            if args[5] == "true":  # the persistent tile walk: nothing in scratch at all (its seams sit inside the stage loop)
                assert k["private_segment_fixed_size"] == 0 and k["vgpr_spill_count"] == 0, (n, k)


def _loop(name):
    import code_object
    from qqq_amd import build

    sym = [k["name"] for k in code_object.kernels(build.LIB) if k["demangled"] == name][0]
    mix = code_object.hottest_loop(build.LIB, sym)
    waits = mix.pop("waits")
    return mix, waits


def _count(mix, prefix, exclude=()):
    return sum(v for o, v in mix.items() if o.startswith(prefix) and not any(o.startswith(e) for e in exclude))


def test_steady_state_loops(table):
    """Instruction mix of the hot loops, from the disassembly (tools/code_object.py::hottest_loop).  What the measured numbers rest
    on and a source or compiler change could silently undo: the unroll (MFMAs per trip), no scratch and no full `vmcnt(0)` drain
    inside a register-ring loop (the waits are counted), the barrier count, and the VALU : MFMA issue ratio of the panel shapes
    (VALU and MFMA share the SIMD's issue port -- 3.1 per MFMA had the loop VALU-issue bound)."""
    # 64 columns per wave, per-channel: 4 stages x (8 m-tiles x 4 operands); barrier after every other stage
    mix, waits = _loop("qqq_panel_kernel<8,false,4,2,4,2,2>")
    assert mix["v_mfma_i32_16x16x64_i8"] == 128 and mix["s_barrier"] == 2
    assert _count(mix, "scratch") == 0 and not any("vmcnt(0)" in w for w in waits)
    assert mix["ds_read_b128"] == 32 and mix["ds_write_b128"] == 8 and mix["global_load_dwordx4"] == 16
    assert _count(mix, "v_", exclude=("v_mfma",)) <= 1.6 * 128
    # ... per-group: the re-quantiser (3 v_pk_fma + v_pk_add + 2 v_and_or + v_perm + v_xor per 4 weights) on top
    mix, waits = _loop("qqq_panel_kernel<8,true,4,2,4,2,2>")
    assert mix["v_mfma_i32_16x16x64_i8"] == 128 and mix["s_barrier"] == 2 and _count(mix, "scratch") == 0
    assert mix["v_pk_fma_f16"] == 192 and mix["v_and_or_b32"] == 128 and not any("vmcnt(0)" in w for w in waits)
    # 32 columns per wave, 128 tokens (the M = 128 point): 4 stages x (8 m-tiles x 2 operands), a barrier per stage
    mix, waits = _loop("qqq_panel_kernel<8,false,4,2,4,2,1>")
    assert mix["v_mfma_i32_16x16x64_i8"] == 64 and mix["s_barrier"] == 4
    assert _count(mix, "scratch") == 0 and not any("vmcnt(0)" in w for w in waits)
    assert _count(mix, "v_", exclude=("v_mfma",)) <= 2.0 * 64
    # the wide kernel (large m since round 3): 4 stages x 2 steps x 64 in-place MFMAs per trip, one barrier per stage, no
    # scratch inside the loop, counted waits, and the issue budget of a LONE wave: at most 4 instructions per MFMA
    # (16 matrix-pipe cycles = 4 issue slots) in the per-channel mode.  LDS-DMA staging: no ds_write at all, per trip
    # 4 x 8 activation DMAs + 8 x 2 ring refills (+ 4 x 2 scale words per-group), every one inline asm.
    # Both ring depths of both modes (the defaults of the 256 x 256 tiles are <false,...,8,...> and <true,...,4,...> since round 5).
    # (first template argument, round 6: 0 per-channel, 1 per-group, 2 = expanded int8 weights -- one 16-byte load per column set and step, no VALU at all)
    for name, mode in (("qqq_wide_kernel<0,16,4,8,2,false>", 0), ("qqq_wide_kernel<1,16,4,4,2,false>", 1),
                       ("qqq_wide_kernel<0,16,4,4,2,false>", 0), ("qqq_wide_kernel<1,16,4,8,2,false>", 1),
                       ("qqq_wide_kernel<2,16,4,4,2,false>", 2)):
        grouped = mode == 1
        mix, waits = _loop(name)
        assert mix["v_mfma_i32_16x16x64_i8"] == 512 and mix["s_barrier"] == 4, name
        assert _count(mix, "scratch") == 0 and not any("vmcnt(0)" in w for w in waits), (name, waits)
        # (round 6, dword weight loads: per trip 32 LDS-DMAs of 16 bytes + 8 steps x 8 one-word ring loads in the packed modes; expanded weights: 32 + 32 sixteen-byte loads)
        assert mix["ds_read_b128"] == 128 and _count(mix, "ds_write") == 0 and mix["buffer_load_dwordx4"] == (64 if mode == 2 else 32), name
        assert mix.get("buffer_load_dword", 0) == (0 if mode == 2 else 64 + (8 if grouped else 0)), name
        assert _count(mix, "v_accvgpr") == 0, name  # accumulators never leave the accumulation registers
        total = sum(v for v in mix.values() if isinstance(v, int))
        assert total <= (4.4 if grouped else 1.75 if mode == 2 else 2.6) * 512, (name, total)
        if mode == 2:  # the loop of the expanded weights: MFMAs, fragment reads, loads and scalar bookkeeping -- no transpose, no unpack, no re-quantiser
            assert _count(mix, "v_", exclude=("v_mfma",)) <= 8, (name, mix)
        assert mix.get("s_nop", 0) <= 72, (name, mix.get("s_nop"))  # hazard fillers: the paired re-quantisations keep them out (round 6, one-instruction transpose items: 59 -> 67 per-group)
    # decode and a-few-tokens kernels: counted waits only, no LDS in the loop, no scratch
    for name in ("qqq_column_kernel<1,false,8,3>", "qqq_column_kernel<1,true,8,3>", "qqq_stream_kernel<1,false,4,3>"):
        mix, waits = _loop(name)
        assert _count(mix, "scratch") == 0 and _count(mix, "ds_") == 0 and mix.get("s_barrier", 0) == 0, name
        assert waits and not any("vmcnt(0)" in w for w in waits), (name, waits)
    # the large-m LDS-DMA tile: 32 x v_mfma_i32_32x32x32_i8 per 128-k block, no scratch
    mix, _ = _loop("qqq_tiled_kernel<256,2,2,2,false,7>")
    assert mix["v_mfma_i32_32x32x32_i8"] == 32 and _count(mix, "scratch") == 0


def test_wide_kernel_hand_counted_waits_replayed_on_the_compiled_code():
    """Every vector-memory load of the wide kernel's loop is inline asm and its waits are hand-counted (LDS-DMA staging: hipcc
    cannot count what it cannot see).  tools/check_waits.py replays the compiled instruction stream -- the loop twice, then every
    feasible path through the ragged tail up to the drain, and the prologue's asm loads -- with loads retiring in issue order, and reports any instruction that
    touches a register a load still in flight is going to write (a too-large count, or hipcc reusing the destination of a dead
    load: both happened while this was written).  All twelve instantiations; and at each barrier exactly the current stage's
    loads may be in flight (the previous stage's DMAs, which the barrier publishes, have retired)."""
    # (round 6, balanced issue slots: the waits moved with the items they belong to -- the replay is what says the new positions are right)
    import check_waits
    import code_object
    from qqq_amd import build

    ks = {k["demangled"]: k["name"] for k in code_object.kernels(build.LIB)}
    for mode in (0, 1, 2):
        grouped = mode == 1
        for mt, hw in ((16, 2), (8, 2), (16, 1)):
            for rs in ((4,) if mode == 2 else (4, 8)):
                name = f"qqq_wide_kernel<{mode},{mt},4,{rs},{hw},false>"
                text = code_object.disassemble(build.LIB, ks[name])
                body, paths = check_waits.tail_paths(text.split("\n"))
                assert sum("v_mfma" in x for x in body) == 16 * hw * mt and len(paths) == 4, (name, len(paths))
                problems, at_barrier = check_waits.check(body)
                for path in paths:
                    problems += check_waits.check(body, path)[0]
                problems += check_waits.check_prologue(text.split("\n"))
                assert not problems, (name, sorted(set(problems))[:4])
                # at the barrier exactly the CURRENT stage's activation chunks may be in flight -- the stage it publishes (issued P - 3 stages earlier) has
                # retired -- next to ring refills / scale loads of the last steps.  (Round 6: the stage's wait + barrier sit in the last plain slot of its second
                # step: in the 128-column shape that is slot 26 of 32, in front of the step's last chunk.)
                chunks = 2 * (mt // 4) - (1 if hw == 1 else 0)
                assert len(at_barrier) == 4 and all(b == at_barrier[0] for b in at_barrier), (name, at_barrier)
                assert at_barrier[0].count("D") == chunks and set(at_barrier[0]) <= set("rD"), (name, at_barrier)
                refills = (2 * hw if mode == 2 else 4 * hw) * 2 + (2 if grouped else 0)  # per stage (packed modes: four one-word loads per half and step)
                assert refills - 1 <= at_barrier[0].count("r") <= refills + 2 * hw, (name, at_barrier)
                # what hipcc's own bookkeeping cannot see around the inline asm (tools/check_vmem.py): M0 is written nowhere but
                # in front of the LDS-DMA that reads it (hipcc reserves M0 -- a clobber is refused as "reserved register" -- so the
                # discipline is checked on the code instead), and no vector-memory instruction reads an SGPR inside the 5 wait
                # states behind a VALU write of it
                import check_vmem

                ins = check_vmem.parse(text)
                assert not check_vmem.m0_discipline(ins, None), name
                assert not check_vmem.sgpr_vmem_hazards(ins), (name, check_vmem.sgpr_vmem_hazards(ins)[:3])


def test_check_waits_flags_planted_faults():
    """The wait replay must catch what it exists for.  A toy loop in compiler syntax: a register load consumed three instructions
    later behind `vmcnt(1)` with one younger load in flight is fine; a count one too large (`vmcnt(2)`) is flagged, and so is
    a value written into the destination of a load still in flight (the dead-load register reuse of the first LDS-DMA build)."""
    import check_waits

    def loop(wait, younger):
        body = [".LBB0_1:", "\tbuffer_load_dwordx4 v[10:13], v1, s[4:7], s8 offen"]
        body += ["\tbuffer_load_dwordx4 v2, s[4:7], s9 offen lds"] * younger
        body += ["\tv_mfma_i32_16x16x64_i8 a[0:3], v[20:23], v[24:27], a[0:3]", f"\ts_waitcnt vmcnt({wait})",
                 "\tv_and_b32_e32 v30, 0xf0f0f0f0, v10", "\ts_waitcnt vmcnt(0)", "\ts_cbranch_scc1 .LBB0_1"]
        return body

    assert check_waits.check(check_waits.loop_body(loop(1, 1)))[0] == []
    assert check_waits.check(check_waits.loop_body(loop(2, 2)))[0] == []
    bad = check_waits.check(check_waits.loop_body(loop(2, 1)))[0]
    assert bad and all("v_and_b32" in x for x in bad), bad
    reuse = [".LBB0_1:", "\tbuffer_load_dwordx4 v[10:13], v1, s[4:7], s8 offen", "\tv_mfma_i32_16x16x64_i8 a[0:3], v[20:23], v[24:27], a[0:3]",
             "\tv_mov_b32_e32 v12, 0", "\ts_waitcnt vmcnt(0)", "\ts_cbranch_scc1 .LBB0_1"]
    bad = check_waits.check(check_waits.loop_body(reuse))[0]
    assert bad and "v_mov_b32" in bad[0], bad



CHAIN_KERNELS = [f"qqq_wide_kernel<{g},{mt},4,{rs},{hw},true>" for g, rs in ((0, 4), (1, 4)) for mt, hw in ((16, 2), (8, 2), (16, 1))] + ["qqq_wide_kernel<2,16,4,4,2,true>"]


def test_tile_walk_waits_replayed_over_the_whole_control_flow_graph():
    """The persistent tile walk (round 4) has a seam behind every stage of its trip: tools/check_vmem.py pushes the queue of loads
    in flight through the kernel's whole control-flow graph to a fixpoint (loads retire in order; stores left out, which only
    makes the replay stricter) and reports every instruction that touches a register a load in flight will write -- a lax
    hand-counted wait, a compiler copy / spill / re-use of a ring register across a seam.  Plus the two hazards hipcc's own
    bookkeeping misses around inline asm: an LDS-DMA without its M0 write, and a vector-memory instruction that reads an SGPR
    a VALU instruction wrote fewer than 5 wait states earlier (stale scalar offsets on the device: the first bug of the walk).
    All six instantiations; the stage loop keeps the plain kernel's instruction budget."""
    import check_vmem
    import code_object
    from qqq_amd import build

    ks = {k["demangled"]: k["name"] for k in code_object.kernels(build.LIB)}
    for name in CHAIN_KERNELS:
        ins = check_vmem.parse(code_object.disassemble(build.LIB, ks[name]))
        problems, info = check_vmem.run(ins)
        real = {i: t for i, t in problems.items() if not check_vmem.scalar_peek(ins, i)}
        assert not real, (name, sorted(real.items())[:4])
        assert not check_vmem.m0_discipline(ins, info), name
        assert not check_vmem.sgpr_vmem_hazards(ins), (name, check_vmem.sgpr_vmem_hazards(ins)[:3])
        args = name.split("<")[1].rstrip(">").split(",")
        mt, hw = int(args[1]), int(args[4])
        loop_mfma = sum(1 for _, op, rest in ins if op.startswith("v_mfma") and not rest.rstrip().endswith(", 0"))
        reset_mfma = sum(1 for _, op, rest in ins if op.startswith("v_mfma") and rest.rstrip().endswith(", 0"))
        assert loop_mfma == 4 * 2 * 2 * hw * mt, (name, loop_mfma)       # four stage copies, no tail copies
        assert reset_mfma == 4 * 2 * hw * mt, (name, reset_mfma)          # one seam per stage position: in-place accumulator resets
        assert not any(op.startswith("scratch_") for _, op, _ in ins), name


def test_check_vmem_flags_planted_faults():
    """The CFG replay must catch what it exists for, on toy kernels in compiler syntax: a consumer behind a wait that is one too
    lax -- also when the load and the consumer sit in different blocks of a loop with a side branch; a copy of a register whose
    load is in flight; a scalar offset written by v_readfirstlane one MFMA ahead of the load that reads it."""
    import check_vmem

    def kernel(wait, extra=()):
        return "\n".join([
            "\ts_mov_b32 s8, 0", ".LBB0_1:", "\tbuffer_load_dwordx4 v[10:13], v1, s[4:7], s8 offen",
            "\tbuffer_load_dwordx4 v2, s[4:7], s9 offen lds", "\ts_cmp_eq_u32 s20, 0", "\ts_cbranch_scc1 .LBB0_3",
            "\tv_mfma_i32_16x16x64_i8 a[0:3], v[20:23], v[24:27], a[0:3]", *extra, ".LBB0_3:", f"\ts_waitcnt vmcnt({wait})",
            "\tv_and_b32_e32 v30, 0xf0f0f0f0, v10", "\ts_waitcnt vmcnt(0)", "\ts_add_i32 s21, s21, -1", "\ts_cmp_lg_u32 s21, 0",
            "\ts_cbranch_scc1 .LBB0_1", "\ts_endpgm"])

    assert check_vmem.run(check_vmem.parse(kernel(1)))[0] == {}
    bad = check_vmem.run(check_vmem.parse(kernel(2)))[0]
    assert bad and all("v_and_b32" in t for t in bad.values()), bad
    bad = check_vmem.run(check_vmem.parse(kernel(1, extra=["\tv_mov_b32_e32 v40, v12"])))[0]
    assert bad and all("v_mov_b32" in t for t in bad.values()), bad
    # stores are not queued: a wait that relies on counting one is flagged (the conservative direction)
    assert check_vmem.run(check_vmem.parse(kernel(2, extra=["\tglobal_store_dword v[50:51], v52, off"])))[0]
    hz = check_vmem.sgpr_vmem_hazards(check_vmem.parse("\n".join([
        "\tv_readfirstlane_b32 s0, v4", "\tv_mfma_i32_16x16x64_i8 a[0:3], v[20:23], v[24:27], a[0:3]", "\ts_nop 0",
        "\tbuffer_load_dwordx4 v[10:13], v1, s[4:7], s0 offen", "\ts_endpgm"])))
    assert len(hz) == 1
    ok = check_vmem.sgpr_vmem_hazards(check_vmem.parse("\n".join([
        "\tv_readfirstlane_b32 s0, v4", "\tv_mfma_i32_16x16x64_i8 a[0:3], v[20:23], v[24:27], a[0:3]", "\ts_nop 4",
        "\tbuffer_load_dwordx4 v[10:13], v1, s[4:7], s0 offen", "\ts_endpgm"])))
    assert ok == []
    # an exit flag the structurizer would leave behind: the path it rules out is not walked
    flagged = "\n".join([
        ".LBB0_1:", "\tbuffer_load_dwordx4 v[10:13], v1, s[4:7], s8 offen", "\ts_mov_b64 s[24:25], 0", "\ts_and_b64 vcc, exec, s[24:25]",
        "\ts_cbranch_vccz .LBB0_9", "\tv_mov_b32_e32 v40, v12", ".LBB0_9:", "\ts_waitcnt vmcnt(0)", "\ts_endpgm"])
    assert check_vmem.run(check_vmem.parse(flagged))[0] == {}


def test_wide_transpose_selects_find_their_vcc_mask():
    """Round 6 (balanced issue slots): the quad transpose's pieces are issued one instruction at a time -- `s_mov_b64 vcc, <lane mask>` in one slot, its two
    `v_cndmask_b32_dpp ... vcc` selects in later ones, MFMAs and other items in between -- so VCC carries a value across asm statements hipcc knows nothing about.
    On the compiled code of every wide-kernel instantiation: each mask write is followed by exactly its two selects before the next one, no select runs without a mask,
    and nothing else writes VCC in between."""
    import re

    import code_object
    from qqq_amd import build

    ks = {k["demangled"]: k["name"] for k in code_object.kernels(build.LIB)}
    names = [n for n in ks if n.startswith("qqq_wide_kernel<") and not n.startswith("qqq_wide_kernel<2,")]  # (expanded weights: no transpose in the loop)
    assert len(names) >= 12
    for name in names:
        pending, selects, problems = 0, 0, []
        for line in code_object.disassemble(build.LIB, ks[name]).split("\n"):
            if not line.startswith("\t"):
                continue
            op, _, args = line.strip().partition(" ")
            first = args.split(",")[0].strip()
            if op == "s_mov_b64" and first == "vcc":
                if pending:
                    problems.append(("mask rewritten with selects outstanding", line))
                pending = 2
            elif op == "v_cndmask_b32_dpp":
                if not pending:
                    problems.append(("select without its mask", line))
                pending = max(0, pending - 1)
                selects += 1
            elif pending and (first == "vcc" or (("_co_" in op or op.startswith("v_div_scale") or op.startswith("v_cmp")) and re.search(r"\bvcc\b", args))):
                problems.append(("VCC written between a mask and its selects", line))
        # (with QQQ_WIDE_DWORD -- the shipped default -- the weights arrive as words and there is no transpose at all: zero selects; a -DQQQ_WIDE_DWORD=0 build has >= 32)
        assert not problems and pending == 0 and (selects == 0 or selects >= 32), (name, problems[:3], pending, selects)


def test_check_waits_replays_lds_reads_too():
    """Round 6: the wide kernel's fragment re-reads are inline asm (ds_read_b128) with hand-placed `s_waitcnt lgkmcnt(N)` -- one per group of four m-tiles, in a slot that
    carries no memory instruction.  tools/check_waits.py keeps a second in-order queue for LDS reads: an MFMA whose operand is the destination of a read that the waits
    so far have not retired is flagged.  Planted faults: a count one too large, and a missing wait; the exact count passes.  Then every plain wide-kernel instantiation
    of the product library, loop + tails (the tile walk keeps hipcc's own waits and passes trivially)."""
    import check_waits
    import code_object
    from qqq_amd import build

    def loop(count):
        body = [".LBB0_1:"]
        body += [f"\tds_read_b128 v[{10 + 4 * i}:{13 + 4 * i}], v1 offset:{1024 * i}" for i in range(4)]
        body += ["\tv_mfma_i32_16x16x64_i8 a[0:3], v[40:43], v[44:47], a[0:3]"]
        if count is not None:
            body += [f"\ts_waitcnt lgkmcnt({count})"]
        body += ["\tv_mfma_i32_16x16x64_i8 a[4:7], v[40:43], v[14:17], a[4:7]", "\ts_waitcnt lgkmcnt(0)", "\ts_cbranch_scc1 .LBB0_1"]
        return body

    assert check_waits.check(check_waits.loop_body(loop(2)))[0] == []          # v[14:17] is the second of four reads: two younger ones may be outstanding
    assert check_waits.check(check_waits.loop_body(loop(3)))[0] != []          # one too many
    assert check_waits.check(check_waits.loop_body(loop(None)))[0] != []       # no wait at all
    ks = {k["demangled"]: k["name"] for k in code_object.kernels(build.LIB)}
    for name, sym in ks.items():
        if not name.startswith("qqq_wide_kernel<") or name.endswith(",true>"):
            continue
        text = code_object.disassemble(build.LIB, sym)
        body, paths = check_waits.tail_paths(text.split("\n"))
        problems = check_waits.check(body)[0]
        for path in paths:
            problems += check_waits.check(body, path)[0]
        assert not [p for p in problems if p.startswith("LDS")], (name, sorted(set(problems))[:3])
        n_asm_waits = sum(1 for l in body if "lgkmcnt" in l)
        assert n_asm_waits >= 8, (name, n_asm_waits)  # (hipcc's own waits while QQQ_WIDE_XWAIT is off: the replay is the same)


def test_wide_loop_issue_slots_stay_balanced():
    """Round 6: a wave that is alone on its SIMD issues an instruction every ~5.2 cycles, and a 16-cycle MFMA hides two of them -- a third in the same slot costs its full issue
    time and no empty slot gives it back (tools/mfma_issue_bench.hip, profiles/r06_mfma_issue_model.txt).  The per-channel loops are scheduled against that: one-instruction
    unpack items dealt to the slots by capacity, the weights as one-word loads (no transpose).  tools/slot_load.py reads the result off the COMPILED loop; a source or compiler
    change that piles instructions into one slot again fails here.  (The per-group loop is over the budget by construction -- 17 VALU per packed word -- and is only held
    to its instruction count.)"""
    import code_object
    import slot_load
    from qqq_amd import build

    ks = {k["demangled"]: k["name"] for k in code_object.kernels(build.LIB)}

    def slots_of(name):
        n, body = slot_load.hottest_loop_lines(build.LIB, ks[name])
        return n, slot_load.slots(body)

    n, sl = slots_of("qqq_wide_kernel<0,16,4,4,2,false>")  # the 256 x 256 tiles of the 4096-token point
    assert n == 512
    heavy = [s for s in sl if len(s) > 2]
    assert len(heavy) <= 1, heavy[:4]  # (the loop's last slot carries the trip counter and the branch)
    assert sum(len(s) for s in sl) <= 1.2 * n, sum(len(s) for s in sl)
    alone = [s for s in sl if "ds_read_b128" in s]
    assert len(alone) == 128 and sum(len(s) > 2 for s in alone) <= 1
    n, sl = slots_of("qqq_wide_kernel<0,8,4,4,2,false>")  # the 128 x 256 tiles of the 1024-token point
    assert n == 256 and sum(len(s) for s in sl) <= 1.95 * n and sum(len(s) > 3 for s in sl) <= 2, (sum(len(s) for s in sl), [s for s in sl if len(s) > 3][:4])
    n, sl = slots_of("qqq_wide_kernel<1,16,4,4,2,false>")  # per-group
    assert n == 512 and sum(len(s) for s in sl) <= 3.0 * n, sum(len(s) for s in sl)
