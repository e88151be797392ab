#!/bin/bash
# rocprofv3 --kernel-trace --stats of tools/prof_f32_engine.py (eager launches: per-kernel rows) -> gpurun_out/<tag>_kernel_stats.csv
# usage: tools/prof_f32.sh <tag> <matmul: f32|bf16x3>
TAG=${1:-f32prof}; MM=${2:-bf16x3}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -- python $REPO/tools/prof_f32_engine.py --matmul $MM --graph 0 --episodes 2 > /tmp/prof_$TAG.out 2>&1 || echo "rocprofv3 failed/timeout"
tail -2 /tmp/prof_$TAG.out
F=$(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1)
mkdir -p $REPO/gpurun_out
cp "$F" $REPO/gpurun_out/${TAG}_kernel_stats.csv
python - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.2f ms over %d launches" % (tot / 1e6, sum(int(r["Calls"]) for r in rows)))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:24]:
    print("%-110s n=%6s avg %8.2f us  %5.1f %%" % (r["Name"][:110], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
