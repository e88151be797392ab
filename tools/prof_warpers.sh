#!/bin/bash
# rocprofv3 kernel stats of the top_k = 40 warper episodes (fused top-k candidate path) -> gpurun_out/<tag>_kernel_stats.csv ; usage: tools/prof_warpers.sh <tag> ["top_k=40 top_p=0.95"]
TAG=${1:-warpers}; CFG=${2:-top_k=40}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -- python $REPO/tools/bench_warpers.py "$CFG" > /tmp/prof_$TAG.out 2>&1 || echo "rocprofv3 failed/timeout"
F=$(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1)
mkdir -p $REPO/gpurun_out
cp "$F" $REPO/gpurun_out/${TAG}_kernel_stats.csv
python - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.2f ms over %d launches" % (tot / 1e6, sum(int(r["Calls"]) for r in rows)))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:16]:
    print("%-110s n=%6s avg %8.2f us  %5.1f %%" % (r["Name"][:110], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
for r in rows:
    if any(k in r["Name"] for k in ("topc", "topk", "lm_head", "sample_reduce")):
        print("  >> %-100s n=%6s avg %8.2f us" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
