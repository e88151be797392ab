#!/bin/bash
# Everything DESIGN.md / bench.py cite for round 4, collected at ONE tree state on one MI355X -> gpurun_out/r04_* (copied to profiles/ afterwards)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
tools/prof_bench.sh r04_bench > gpurun_out/r04_bench_kernel_top.txt 2>&1
bash tools/pmc_fetch_write.sh > gpurun_out/r04_pmc_top.txt 2>&1; cp gpurun_out/pmc_fetch_write.json gpurun_out/r04_pmc_fetch_write.json
# the bench line LAST of the three: it reports roofline.traffic only from a PMC summary collected on the live kernel sources, and cites the kernel stats
cp gpurun_out/r04_pmc_fetch_write.json gpurun_out/r04_bench_kernel_stats.csv profiles/ 2>/dev/null
python bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err
tools/pmc_mfma_bench.sh > /dev/null 2>&1; cp gpurun_out/pmc_mfma_bench.txt gpurun_out/r04_pmc_mfma_bench.txt
tools/prof_f32.sh r04_bf16x3_rollout bf16x3 > gpurun_out/r04_bf16x3_top.txt 2>&1
tools/prof_f32.sh r04_f32_rollout f32 > gpurun_out/r04_f32_top.txt 2>&1
python tools/ab_rollout_variants.py 0 201 203 2>&1 | grep variant > gpurun_out/r04_ab_rollout_variants.txt
python tools/bench_twentyq_dual.py 2>&1 | grep -v "Warn\|amdgpu" | tail -3 > gpurun_out/r04_twentyq_dual_model_rollout.txt
tail -c 300 gpurun_out/r04_bench_default.err; python - <<'PY'
import json
d = json.load(open("gpurun_out/r04_bench_default.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("traffic"), d["fp32_mode"]["value"], d["bf16x3_mode"]["value"], {k: v["ms_per_step"] for k, v in d["train_step"].items()}, d["env_only"]["envs"])
PY
cat gpurun_out/r04_pmc_mfma_bench.txt | head -12
