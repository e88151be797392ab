#!/bin/bash
# rocprofv3 kernel stats + FETCH/WRITE counters of the RL-reduction kernels (bench.py --mode-rl-reduce-only); summaries -> gpurun_out/
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
python $REPO/bench.py --mode-rl-reduce-only > $OUT/r06_rl_reduce_live.json 2> /tmp/rl_live.err || tail -5 /tmp/rl_live.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_rl -- python $REPO/bench.py --mode-rl-reduce-only > /tmp/prof_rl.out 2>&1 || echo "rocprofv3 failed/timeout"
f=$(find /tmp/prof_rl -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/r06_rl_reduce_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "chain_scan|whiten" --output-format csv -d /tmp/pmc_rl_$c -- python $REPO/bench.py --mode-rl-reduce-only > /dev/null 2>&1 || echo "pmc $c failed"
done
python - <<PY
import csv, glob, json, collections
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = glob.glob(f"/tmp/pmc_rl_{c}/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(list)
    for f in fs:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == c:
                acc[(r["Kernel_Name"].split("(")[0][:80], r.get("Grid_Size"))].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        out.setdefault(f"{k[0]} grid={k[1]}", {})[c.lower() + "_kb_avg"] = sum(v) / len(v)
        out[f"{k[0]} grid={k[1]}"]["launches"] = len(v)
json.dump(out, open("$OUT/r06_rl_reduce_pmc.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
PY
cat $OUT/r06_rl_reduce_live.json | python -c "
import json,sys
d=json.load(sys.stdin)['rl_reduce']['chains']
for b,r in d.items():
    print(b, {k:(v['avg_launch_us'], v['achieved'], v['frac']) for k,v in r.items() if isinstance(v,dict)})"
head -12 $OUT/r06_rl_reduce_kernel_stats.csv
