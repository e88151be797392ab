#!/bin/bash
# rocprofv3 --kernel-trace --stats of an arbitrary python command -> gpurun_out/<tag>_kernel_stats.csv + a top-N table (kernel time vs the command's own wall clock)
# usage: tools/prof_any.sh <tag> <script.py> [args...]
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -- python $REPO/"$@" > /tmp/prof_$TAG.out 2>&1 || echo "rocprofv3 failed/timeout"
grep -v "Warning\|amdgpu.ids\|rocprofv3" /tmp/prof_$TAG.out | tail -6
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
PY
