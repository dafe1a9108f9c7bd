#!/usr/bin/env python3
"""From a rocprofv3 kernel trace of bench.py: the proposal chain of the last full step on the main stream -- wall time between the end of the
RPN head's GEMM and the start of RoIAlign forward, the kernel time inside that window per queue, and the launch count."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
roi = [i for i, r in enumerate(rows) if "roi_align_fwd" in r["Kernel_Name"]]
i_roi = roi[-2] if len(roi) > 1 else roi[-1]
q = rows[i_roi]["Queue_Id"]
# the last 3x3 GEMM with bias pass before it on the same queue = RPN head: walk back to the previous nms_scan, then to the p8 kernel before it
j = i_roi
while j > 0 and "nms_scan" not in rows[j]["Kernel_Name"]:
    j -= 1
k = j
while k > 0 and not ("conv_gemm_p8_kernel" in rows[k]["Kernel_Name"] and rows[k]["Queue_Id"] == q):
    k -= 1
t0, t1 = rows[k]["e"], rows[i_roi]["s"]
print(f"window {(t1 - t0) / 1e3:.1f} us from end of {rows[k]['Kernel_Name'][:50]} to start of RoIAlign fwd")
per_q = collections.defaultdict(lambda: [0, 0])
names = collections.Counter()
counts = collections.Counter()
for r in rows[k + 1:i_roi]:
    if r["s"] >= t0 and r["e"] <= t1:
        per_q[r["Queue_Id"]][0] += 1
        per_q[r["Queue_Id"]][1] += r["e"] - r["s"]
        if r["Queue_Id"] == q:
            names[r["Kernel_Name"][:110]] += (r["e"] - r["s"]) / 1e3
            counts[r["Kernel_Name"][:110]] += 1
for qq, (n, ns) in per_q.items():
    print(f"queue {qq}{' (main)' if qq == q else ''}: {n} kernels, {ns / 1e3:.1f} us busy")
for nm, us in names.most_common(40):
    print(f"  {us:8.1f} us  x{counts[nm]:3d}  {nm}")
