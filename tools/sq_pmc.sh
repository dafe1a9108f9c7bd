#!/bin/bash
# SQ wave-state counters for the RoIAlign kernels (run on the GPU box):  bash tools/sq_pmc.sh <tag>
# One pass of 8 SQ counters, kernel-trace only (no other trace domains), on tools/pmc_target.py.
set -u
tag=${1:-r2}
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_sq
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d /tmp/pmc_sq -o p -- python3 $GRAFT_REPO_ROOT/tools/pmc_target.py > /tmp/pmc_sq.log 2>&1
f=$(find /tmp/pmc_sq -name "*counter_collection.csv" | head -1)
if [ -n "$f" ]; then
  python3 - "$f" "$out/pmc_sq.csv" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"][:70]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(k, r["Counter_Name"])] += 1
with open(sys.argv[2], "w") as f:
    names = sorted({c for v in agg.values() for c in v})
    f.write("kernel," + ",".join(names) + ",dispatches\n")
    for k, v in agg.items():
        n = max(cnt[(k, c)] for c in names)
        f.write(k.replace(",", ";") + "," + ",".join(f"{v[c] / max(cnt[(k, c)], 1):.0f}" for c in names) + f",{n}\n")
print(open(sys.argv[2]).read())
PY
else echo "no counter csv"; tail -5 /tmp/pmc_sq.log; fi
