#!/usr/bin/env python3
"""Per-kernel means of the counters collected by tools/pmc_kernel.sh (rocprofv3 --pmc, one csv per pass) + derived lines."""
import collections, csv, glob, os, sys
d = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(d, "p*", "**", "*counter_collection.csv"), recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        name = name[5:] if name.startswith("void ") else name
        k = (name.split("(")[0][:90], r["Grid_Size"], r.get("Workgroup_Size", ""))
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, cs in sorted(agg.items()):
    if "qqq_" not in k[0] or "dynamic_quant" in k[0] or "pack" in k[0]:
        continue
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    print(f"== {k[0]}  grid={k[1]} wg={k[2]}  launches={len(next(iter(cs.values())))}")
    if dur.get(k):
        v = dur[k]
        print(f"   duration under the profiler: mean {sum(v)/len(v):.1f} us  min {min(v):.1f}  (n={len(v)})")
    for c in sorted(m):
        print(f"   {c:28s} {m[c]:.4e}")
    g = m.get("GRBM_GUI_ACTIVE")
    wc = m.get("SQ_WAVE_CYCLES")
    if g and dur.get(k):
        # GRBM_GUI_ACTIVE counts from before the first wave to after the last one (command processor, cache write-back): for a
        # launch of a few tens of microseconds that window is much longer than the kernel's own timestamps, and the ratio came
        # out ABOVE the part's 2.4 GHz maximum (2.7 - 7.6 "GHz" for the panel / stream / column / reduce kernels in round 3).
        # It is a usable clock estimate only for long launches, and never above the maximum.
        mean_us = sum(dur[k]) / len(dur[k])
        ghz = g / 8 / mean_us / 1e3
        if mean_us >= 100.0 and ghz <= 2.4:
            print(f"   -> effective clock ~ {ghz:.2f} GHz (GRBM_GUI_ACTIVE / 8 XCDs / duration)")
        else:
            print(f"   -> effective clock: n/a (launch of {mean_us:.0f} us: the GRBM_GUI_ACTIVE window is not the kernel's; bench.py `clocks` has rocm-smi's reading)")
    if wc:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_VALU"):
            if c in m:
                print(f"   -> {c} / SQ_WAVE_CYCLES = {100 * m[c] / wc:.1f} %")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and g:
        # busy cycles are summed over SIMDs; 1024 SIMDs; GRBM_GUI_ACTIVE summed over 8 XCDs
        print(f"   -> matrix pipe busy = {100 * m['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * g / 8):.1f} % of SIMD cycles"
              f"  ({m['SQ_VALU_MFMA_BUSY_CYCLES'] / max(m.get('SQ_INSTS_MFMA', 1), 1):.1f} cycles per MFMA)")
    if "SQ_LDS_BANK_CONFLICT" in m and m.get("SQ_LDS_IDX_ACTIVE"):
        print(f"   -> LDS bank conflicts = {100 * m['SQ_LDS_BANK_CONFLICT'] / m['SQ_LDS_IDX_ACTIVE']:.1f} % of LDS-active cycles;"
              f" LDS busy = {100 * m['SQ_LDS_IDX_ACTIVE'] / (256 * g / 8):.1f} % of CU cycles" if g else "")
    if "FETCH_SIZE" in m:
        print(f"   -> fabric read  = {2 * m['FETCH_SIZE'] * 1024 / 1e6:.1f} MB per launch (2 x FETCH_SIZE KiB, gfx950 correction)")
    if "WRITE_SIZE" in m:
        print(f"   -> fabric write = {m['WRITE_SIZE'] * 1024 / 1e6:.1f} MB per launch")
