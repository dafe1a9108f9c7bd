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
ap.add_argument("--sgd-per-step", type=int, default=1, help="fused optimizer launches per training step (targetDET: 2 -- student + CKG)")
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

# ---- per-category device time and per-queue occupancy (round 6: the account of the targetDET step)
import re

CATS = [("conv GEMM fwd + dgrad (hand-written)", r"conv_gemm_p8_kernel|conv_gemm_s4_kernel|conv_gemm_bf16_kernel|conv_gemm256|p8_slab_sum|p8_tail|s4_tail"),
        ("weight gradient + its reductions (hand-written)", r"conv_wgrad|tn_reduce|wgrad_reduce|wgrad_s4"),
        ("BatchNorm / pool streams (hand-written)", r"bn_apply|bn_bwd|bn_stats|bn_finalize|conv_stats_finalize|avgpool2"),
        ("RoIAlign", r"roi_align"),
        ("NMS + labelling + samplers (hand-written)", r"nms_|anchor_match|sample_labels|lowq|match_"),
        ("box head GEMMs / losses / SGD / EMA (hand-written)", r"gemm_nt_bf16|bias_act|cosine_|mil_|kl_div|box_reg|l1_mean|rpn_losses|sgd_kernel|ema_kernel|weight_dgrad|normalize_pad|transpose_kernel|window_attn|aug_"),
        ("library GEMMs + attention (hipBLASLt / SDPA)", r"^Cijk_|attn_fwd|attn_bwd|bwd_kernel|fmha|flash"),
        ("library convolutions (MIOpen / CK)", r"igemm|ck::|kernel_grouped_conv|naive_conv|miopen|Conv|gridwise"),
        ("runtime copies / fills", r"__amd_rocclr|fillBuffer|copyBuffer"),
        ("torch elementwise / reduce / sort / index glue", r"at::native|at_cuda_detail|rocprim|hipcub|elementwise|reduce_kernel|sort|scan")]
cat_t = collections.OrderedDict((c, [0, 0]) for c, _ in CATS)
cat_t["other"] = [0, 0]
for n, (c, t) in agg.items():
    for name, rx in CATS:
        if re.search(rx, n):
            cat_t[name][0] += c
            cat_t[name][1] += t
            break
    else:
        cat_t["other"][0] += c
        cat_t["other"][1] += t
nsteps = max(1, sum(c for n, (c, t) in agg.items() if "sgd_kernel" in n) // a.sgd_per_step)   # fused optimizer launches mark the steps
print(f"-- device time by category over the window ({a.window_ms} ms = {nsteps} training steps: {a.window_ms / nsteps:.2f} ms per step under the profiler; "
      "kernels of concurrent streams add up to more than the wall time)")
for name, (c, t) in cat_t.items():
    print(f"{t/1e6:8.2f} ms {100.0*t/max(busy,1):5.1f}% n={c:6d}  | per step {t/1e6/nsteps:6.2f} ms, {c/nsteps:6.1f} launches  {name}")
# per hardware queue: time during which that queue has a kernel running (union of its intervals)
qiv = collections.defaultdict(list)
for (s0, e0, _), q in zip(rows, queues):
    if e0 > lo:
        qiv[q].append((max(s0, lo), e0))
print("-- per hardware queue: occupied ms (union of its kernels) / kernel count")
for q, iv2 in sorted(qiv.items(), key=lambda kv: -sum(e - s for s, e in kv[1]))[:8]:
    iv2.sort()
    u, cs, ce = 0, None, None
    for s0, e0 in iv2:
        if ce is None or s0 > ce:
            if ce is not None:
                u += ce - cs
            cs, ce = s0, e0
        else:
            ce = max(ce, e0)
    if ce is not None:
        u += ce - cs
    print(f"   q{q}: {u/1e6:8.1f} ms occupied = {100.0*u/(a.window_ms*1e6):5.1f}% of the window, {len(iv2)} kernels")

# ---- gaps between consecutive kernels of the busiest queue (late round 6): how much of a step is the queue's own launch-to-launch dead time?
# A gap of a few microseconds after a kernel is the device's (drain, dependency resolution, dispatch ramp -- inside a HIP-graph replay the host
# is not involved); gaps of tens of microseconds and more are the host (under the profiler more than without it) or a cross-stream wait.
if qiv:
    qmain = max(qiv.items(), key=lambda kv: sum(e - s for s, e in kv[1]))[0]
    seq = sorted((s0, e0, n) for (s0, e0, n), q in zip(rows, queues) if q == qmain and e0 > lo)
    edges = [0.5, 1, 2, 3, 4, 6, 10, 20, 50, 1e9]
    cnt, tot = [0] * len(edges), [0.0] * len(edges)
    after = collections.defaultdict(lambda: [0, 0.0])
    for (s0, e0, n0), (s1, e1, n1) in zip(seq, seq[1:]):
        g = (s1 - e0) / 1e3
        if g <= 0:
            continue
        k = next(i for i, x in enumerate(edges) if g <= x)
        cnt[k] += 1
        tot[k] += g
        if g <= 10:
            key = n0.split("(")[0].split("<")[0][-60:]
            after[key][0] += 1
            after[key][1] += g
    print(f"-- gaps between consecutive kernels of queue q{qmain} in the window ({len(seq)} kernels, {nsteps} steps): count / total ms / ms per step, by gap length")
    lo_e = 0
    for x, c, t in zip(edges, cnt, tot):
        print(f"   {lo_e:5g} - {x if x < 1e9 else float('inf'):5g} us: {c:6d} gaps {t/1e3:8.2f} ms  {t/1e3/nsteps:6.2f} ms/step")
        lo_e = x
    small = sum(t for x, t in zip(edges, tot) if x <= 10)
    print(f"   gaps <= 10 us: {small/1e3/nsteps:.2f} ms per step over {sum(c for x, c in zip(edges, cnt) if x <= 10)/nsteps:.0f} kernel boundaries per step")
    print("   ... of which, by the kernel BEFORE the gap (top 12 by total):")
    for key, (c, t) in sorted(after.items(), key=lambda kv: -kv[1][1])[:12]:
        print(f"      {t/1e3/nsteps:6.3f} ms/step  n/step={c/nsteps:6.1f}  mean {t/c:5.2f} us  {key}")

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
