#!/bin/bash
# MFMA-pipe busy fraction of the fp32 train step's kernels (is the f32 sgemm's 0.72 of nominal peak issue-bound or clock-bound?) -> gpurun_out/pmc_mfma_sgemm.txt
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_mfma
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_mfma -- python $REPO/bench.py --mode ilql-step --train-matmul f32 --steps 1 --warmup 1 > /tmp/pmc_mfma.log 2>&1
f=$(find /tmp/pmc_mfma -name "*counter_collection.csv" | head -1)
python - $f <<'PY' > $REPO/gpurun_out/pmc_mfma_sgemm.txt
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][-60:]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE": cnt[k] += 1
for k, n in sorted(cnt.items(), key=lambda kv: -agg[kv[0]]["GRBM_GUI_ACTIVE"])[:8]:
    a = agg[k]
    print("%-60s n=%4d gui_active/launch %10.0f  mfma_busy %14.0f  busy_cu %14.0f  mfma_busy/busy_cu %.3f  mfma_busy/(gui*256cu*4simd?) %.3f  mops_f32 %14.0f" %
          (k, n, a["GRBM_GUI_ACTIVE"] / n, a["SQ_VALU_MFMA_BUSY_CYCLES"], a["SQ_BUSY_CU_CYCLES"], a["SQ_VALU_MFMA_BUSY_CYCLES"] / max(a["SQ_BUSY_CU_CYCLES"], 1),
           a["SQ_VALU_MFMA_BUSY_CYCLES"] / max(a["GRBM_GUI_ACTIVE"] * 256 * 4, 1), a["SQ_INSTS_VALU_MFMA_MOPS_F32"]))
PY
cat $REPO/gpurun_out/pmc_mfma_sgemm.txt; tail -3 /tmp/pmc_mfma.log | cut -c1-300
