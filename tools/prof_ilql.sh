#!/bin/bash
# rocprofv3 kernel stats of the ILQL value-policy rollouts (tools/bench_ilql_rollout.py) -> gpurun_out/<tag>_kernel_stats.csv
TAG=${1:-ilql}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -- python $REPO/tools/bench_ilql_rollout.py > /tmp/prof_$TAG.out 2>&1 || echo "rocprofv3 failed/timeout"
F=$(find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -1)
mkdir -p $REPO/gpurun_out
cp "$F" $REPO/gpurun_out/${TAG}_kernel_stats.csv
python - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.2f ms over %d launches" % (tot / 1e6, sum(int(r["Calls"]) for r in rows)))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:14]:
    print("%-110s n=%6s avg %8.2f us  %5.1f %%" % (r["Name"][:110], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
