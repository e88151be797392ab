#!/bin/bash
# Where the train step's flash-attention kernels spend their wave time (tools/bench_flash_train.py under rocprofv3 --pmc, two passes):
#   wave-time shares  WAIT_ANY (parked: s_waitcnt / barrier) | WAIT_INST_ANY (issue stall) | ACTIVE_INST_ANY, and of the issue slots VALU / LDS / MFMA-busy
# -> gpurun_out/pmc_flash.txt
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_fl1 /tmp/pmc_fl2
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d /tmp/pmc_fl1 -- python $REPO/tools/bench_flash_train.py > /tmp/pmc_fl1.log 2>&1 || echo "pass 1 failed"
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d /tmp/pmc_fl2 -- python $REPO/tools/bench_flash_train.py > /tmp/pmc_fl2.log 2>&1 || echo "pass 2 failed"
python - $(find /tmp/pmc_fl1 -name "*counter_collection.csv" | head -1) $(find /tmp/pmc_fl2 -name "*counter_collection.csv" | head -1) <<'PY' > $REPO/gpurun_out/pmc_flash.txt
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in sys.argv[1:]:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-48:]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] in ("SQ_WAVE_CYCLES",): cnt[k] += 1
print("# tools/bench_flash_train.py (B=32 H=12, T=512 and 1024 together, bf16 and fp32 kernels separately named)")
for k in sorted(agg, key=lambda k: -agg[k]["SQ_WAVE_CYCLES"]):
    a = agg[k]
    w = max(a["SQ_WAVE_CYCLES"], 1)
    print("%-48s wait_any %.2f  wait_inst %.2f  active %.2f | of wave cycles: valu %.3f lds %.3f wait_inst_lds %.3f | mfma_busy/(4 busy_cu) %.3f"
          % (k, a["SQ_WAIT_ANY"] / w, a["SQ_WAIT_INST_ANY"] / w, a["SQ_ACTIVE_INST_ANY"] / w, a["SQ_ACTIVE_INST_VALU"] / w, a["SQ_ACTIVE_INST_LDS"] / w,
             a["SQ_WAIT_INST_LDS"] / w, a["SQ_VALU_MFMA_BUSY_CYCLES"] / max(4 * a["SQ_BUSY_CU_CYCLES"], 1)))
PY
cat $REPO/gpurun_out/pmc_flash.txt; tail -3 /tmp/pmc_fl1.log
