#!/bin/bash
# LDS bank-conflict counters of the two dW kernels (tools/bench_dw_kmajor.py) -> gpurun_out/pmc_lds_dw.txt
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_lds
timeout 500 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pmc_lds -- python $REPO/tools/bench_dw_kmajor.py > /tmp/pmc_lds.log 2>&1
f=$(find /tmp/pmc_lds -name "*counter_collection.csv" | head -1)
python - $f <<'PY' > $REPO/gpurun_out/pmc_lds_dw.txt
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][-70:]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_INSTS_LDS": cnt[k] += 1
for k, n in cnt.most_common(8):
    a = agg[k]
    print("%-70s n=%4d  lds insts/launch %10.0f  bank-conflict cycles/launch %10.0f  idx-active %10.0f  conflict/active %.3f" %
          (k, n, a["SQ_INSTS_LDS"] / n, a["SQ_LDS_BANK_CONFLICT"] / n, a["SQ_LDS_IDX_ACTIVE"] / n, a["SQ_LDS_BANK_CONFLICT"] / max(a["SQ_LDS_IDX_ACTIVE"], 1)))
PY
cat $REPO/gpurun_out/pmc_lds_dw.txt
