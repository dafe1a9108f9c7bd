#!/usr/bin/env python3
"""Reduce a rocprofv3 --pmc counter_collection CSV to per-kernel means for our kernels."""
import csv, sys, collections
src, out = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
with open(src) as f:
    for row in csv.DictReader(f):
        n = row["Kernel_Name"]
        if any(k in n for k in ("roi_align", "bn_", "avgpool2", "gemm_nt", "nms_", "conv_gemm", "conv_wgrad", "wgrad_reduce", "conv_stats")):
            agg[n][row["Counter_Name"]].append(float(row["Counter_Value"]))
with open(out, "w") as f:
    w = csv.writer(f)
    w.writerow(["Kernel_Name", "Counter_Name", "launches", "mean_value"])
    for n, d in agg.items():
        for c, v in d.items():
            w.writerow([n[:120], c, len(v), sum(v) / len(v)])
            print(n[:70], c, len(v), sum(v) / len(v))
