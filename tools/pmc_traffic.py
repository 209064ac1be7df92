#!/usr/bin/env python3
"""Summarise gpurun_out/pmc_{FETCH_SIZE,WRITE_SIZE}/t_counter_collection.csv (tools/pmc_traffic.sh) into
profiles/hbm_traffic.json: bytes per launch = 2 x FETCH_SIZE KiB (gfx950 correction, calibrated in
profiles/r01_hbm_traffic.txt) + WRITE_SIZE KiB."""
import collections, csv, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
vals = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(os.path.join(ROOT, "gpurun_out", f"pmc_{c}", "t_counter_collection.csv"))):
        if r["Counter_Name"] == c and r["Kernel_Name"].startswith("_Z1"):
            agg[(r["Kernel_Name"].split("EEv")[0], r["Grid_Size"])].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        vals.setdefault(k, {})[c] = sum(v) / len(v)
out = {"_comment": "HBM/fabric bytes per launch from rocprofv3 --pmc (separate passes FETCH_SIZE, WRITE_SIZE; tools/pmc_traffic.sh), "
                   "gfx950 correction read = 2 x FETCH_SIZE x 1024 (MI355X_MICROARCH.md HBM section), calibrated on "
                   "qqq_dynamic_quant_kernel; see profiles/r01_hbm_traffic.txt. Counters are L2<->fabric traffic and include Infinity-Cache hits."}
for (name, grid), v in sorted(vals.items()):
    print(name, grid, v)
    key = None
    if "qqq_stream_kernelILi1E" in name: key = "qqq_stream_kernel_M16"
    if "qqq_column_kernelILi1E" in name: key = "qqq_column_kernel_M1"
    if "qqq_tiled_kernel" in name: key = "qqq_tiled_kernel_M4096"
    if "qqq_panel_kernel" in name: key = "qqq_panel_kernel_M4096"
    if key and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        out[key] = {"kernel": name, "fetch_size_kib": round(v["FETCH_SIZE"]), "write_size_kib": round(v["WRITE_SIZE"]),
                    "bytes": int(2 * v["FETCH_SIZE"] * 1024 + v["WRITE_SIZE"] * 1024)}
json.dump(out, open(os.path.join(ROOT, "profiles", "hbm_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
