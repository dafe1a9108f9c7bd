#!/usr/bin/env python3
"""Summarise the steady-state tail of a rocprofv3 --kernel-trace CSV (per-dispatch rows) into a small per-kernel table.

    python tools/trace_summary.py <kernel_trace.csv> <out.csv> [--window-ms 300]

rocprofv3's own --stats aggregates the whole process, i.e. including MIOpen's one-off solver search and the warm-up
steps; the bench's timed steps are the LAST thing the process does, so the last `window-ms` of device time are taken.
"""
import argparse
import csv
import collections

ap = argparse.ArgumentParser()
ap.add_argument("trace")
ap.add_argument("out")
ap.add_argument("--window-ms", type=float, default=300.0)
ap.add_argument("--list", default=None, help="also list every dispatch whose name contains this string inside the last --list-ms of the trace")
ap.add_argument("--list-ms", type=float, default=45.0)
a = ap.parse_args()
rows = []
queues = []
with open(a.trace) as f:
    r = csv.DictReader(f)
    for row in r:
        rows.append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]), row["Kernel_Name"]))
        queues.append(row.get("Queue_Id", "?"))
end = max(e for _, e, _ in rows)
lo = end - int(a.window_ms * 1e6)
agg = collections.defaultdict(lambda: [0, 0])
busy = 0
for s, e, n in rows:
    if s >= lo:
        agg[n][0] += 1
        agg[n][1] += e - s
        busy += e - s
items = sorted(agg.items(), key=lambda kv: -kv[1][1])
with open(a.out, "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "PercentageOfBusy", f"window_ms={a.window_ms}", f"busy_ms={busy/1e6:.3f}"])
    for n, (c, t) in items:
        w.writerow([n[:160], c, t, t // c, f"{100.0*t/busy:.2f}"])
print(f"window {a.window_ms} ms: busy {busy/1e6:.1f} ms over {len(items)} kernels")
# time during which at least one kernel runs (overlapping streams counted once), the idle remainder, and the share per hardware queue
iv = sorted((max(s, lo), e) for s, e, _ in rows if e > lo)
union, cur_s, cur_e = 0, None, None
for s, e in iv:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            union += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
if cur_e is not None:
    union += cur_e - cur_s
perq = collections.defaultdict(int)
for (s, e, _), q in zip(rows, queues):
    if s >= lo:
        perq[q] += e - s
print(f"device occupied (union over streams) {union/1e6:.1f} ms = {100.0*union/(a.window_ms*1e6):.1f}% of the window; idle {a.window_ms - union/1e6:.1f} ms; "
      f"kernel time per hardware queue: " + ", ".join(f"q{q}: {t/1e6:.1f} ms" for q, t in sorted(perq.items(), key=lambda kv: -kv[1])[:6]))
for n, (c, t) in items[:40]:
    print(f"{t/1e6:8.2f} ms {100.0*t/busy:5.1f}% n={c:5d} avg={t/c/1e3:9.1f}us {n[:100]}")

if a.list:
    # each matching dispatch of the last `list-ms`: start offset, duration, and how much OTHER kernel time overlaps it (side streams)
    lo2 = end - int(a.list_ms * 1e6)
    sel = [(s, e, n) for s, e, n in rows if s >= lo2]
    print(f"-- dispatches containing '{a.list}' in the last {a.list_ms} ms (offset ms, duration us, overlapped-by-others us, name)")
    for s0, e0, n in sorted(sel):
        if a.list not in n:
            continue
        ov = sum(max(0, min(e0, e1) - max(s0, s1)) for s1, e1, n1 in sel if (s1, e1, n1) != (s0, e0, n))
        print(f"{(s0 - lo2) / 1e6:8.3f} {(e0 - s0) / 1e3:9.1f} {ov / 1e3:9.1f}  {n[:90]}")
