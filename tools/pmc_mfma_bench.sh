#!/bin/bash
# MFMA-pipe busy fraction per kernel of the rollout bench (eager launches) -> gpurun_out/pmc_mfma_bench.txt
# busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES): the share of the CU-busy time in which a SIMD's matrix pipe is occupied
# (the same normalisation as tools/pmc_mfma_sgemm.sh: the f32 sgemm reads 0.72 - 0.77 there at 0.73 of its nominal TFLOP/s).
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_mfma_b
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc_mfma_b -- python $REPO/bench.py --graph 0 --steps 1 --warmup 1 --no-cpu-baseline --no-train-step --no-fp32-mode --no-ppo-iteration --no-rl-reduce --no-maze > /tmp/pmc_mfma_b.log 2>&1 || echo "pmc pass failed/timeout"
f=$(find /tmp/pmc_mfma_b -name "*counter_collection.csv" | head -1)
python - $f <<'PY' > $REPO/gpurun_out/pmc_mfma_bench.txt
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0][-64:]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE": cnt[k] += 1
print("# bench.py --graph 0 --steps 1 --warmup 1 (eager): MFMA-pipe busy per kernel; busy = MFMA_BUSY_CYCLES / (4 x BUSY_CU_CYCLES)")
for k, n in sorted(cnt.items(), key=lambda kv: -agg[kv[0]]["GRBM_GUI_ACTIVE"])[:14]:
    a = agg[k]
    print("%-64s n=%5d  mfma_busy/(4 busy_cu) %.3f   gui_active per launch %9.0f" % (k, n, a["SQ_VALU_MFMA_BUSY_CYCLES"] / max(4 * a["SQ_BUSY_CU_CYCLES"], 1), a["GRBM_GUI_ACTIVE"] / n))
PY
cat $REPO/gpurun_out/pmc_mfma_bench.txt
