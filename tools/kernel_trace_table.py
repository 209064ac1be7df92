#!/usr/bin/env python3
"""Per-(kernel, launch geometry) duration table from a rocprofv3 --kernel-trace CSV: every sweep point of bench.py gets its
own row (rocprofv3's own --stats lumps all launches of a template instance together).
usage: tools/kernel_trace_table.py <dir with *_kernel_trace.csv> > profiles/rNN_bench_kernels_by_shape.csv"""
import collections, csv, glob, os, sys

rows = collections.defaultdict(list)
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "qqq_" not in name:
            continue
        name = name[5:] if name.startswith("void ") else name
        name = name.split("(")[0]
        if name.startswith("_Z"):  # rocprofv3 leaves some template instances mangled
            sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
            from code_object import _demangle
            name = _demangle(name)
        grid = (int(r["Grid_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
        wg = (int(r["Workgroup_Size_X"]), int(r["Workgroup_Size_Y"]), int(r["Workgroup_Size_Z"]))
        wgs = tuple(g // max(w, 1) for g, w in zip(grid, wg))
        rows[(name, wgs, wg[0], r["LDS_Block_Size"], r["VGPR_Count"], r.get("Accum_VGPR_Count", ""))].append(
            (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
w = csv.writer(sys.stdout)
w.writerow(["kernel", "workgroups_xyz", "threads", "lds_bytes", "vgprs", "agprs", "calls", "avg_us", "min_us", "median_us", "max_us", "total_us"])
for k, v in sorted(rows.items(), key=lambda kv: -sum(kv[1])):
    v = sorted(v)
    w.writerow([k[0], "x".join(map(str, k[1])), k[2], k[3], k[4], k[5], len(v), f"{sum(v)/len(v):.2f}", f"{v[0]:.2f}", f"{v[len(v)//2]:.2f}", f"{v[-1]:.2f}", f"{sum(v):.1f}"])
