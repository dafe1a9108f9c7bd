#!/usr/bin/env python3
"""Per-kernel summary of tools/lab_prof.sh's passes: mean duration per (kernel, grid) from the trace, mean counter values from the PMC passes."""
import collections, csv, glob, os, sys

root = sys.argv[1]


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n[:n.index("(")] if "(" in n else n[:60]


dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        key = (short(r["Kernel_Name"]), r.get("Grid_Size", r.get("Grid_Size_X", "")), r.get("VGPR_Count", ""), r.get("LDS_Block_Size", ""))
        dur[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("== kernel trace: name, grid, vgpr, lds : launches, mean us, min us")
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k[0]:60s} grid={k[1]:>8s} vgpr={k[2]:>4s} lds={k[3]:>7s} n={len(v):4d} mean={sum(v)/len(v):9.1f} min={min(v):9.1f}")
for sub in ("sq", "fetch", "write"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(root, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            agg[(short(r["Kernel_Name"]), r.get("Grid_Size", ""))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f"== pmc pass {sub}: mean per launch")
    for k, d in sorted(agg.items()):
        print(f"{k[0]:60s} grid={k[1]:>8s} " + "  ".join(f"{c}={sum(v)/len(v):.4g}" for c, v in sorted(d.items())))
