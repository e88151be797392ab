#!/bin/bash
# per-kernel durations of tools/bench_flash_train.py (default kernels and the round-3 ones) -> gpurun_out/flash_kernel_stats.txt
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/flprof
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/flprof -- python $REPO/tools/bench_flash_train.py --variants 0 7 > /tmp/flprof.log 2>&1 || { echo "rocprofv3 failed"; tail -5 /tmp/flprof.log; }
f=$(find /tmp/flprof -name "*kernel_stats.csv" | head -1)
python - $f <<'PY' | tee $REPO/gpurun_out/flash_kernel_stats.txt
import csv, sys
print("# kernel, calls, avg us, min us (T = 512 and T = 1024 launches pooled: min = T 512)")
for r in csv.DictReader(open(sys.argv[1])):
    if "flash" in r["Name"]:
        print("%-72s %4s  avg %8.1f  min %8.1f  max %8.1f" % (r["Name"].split("(")[0][-72:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
