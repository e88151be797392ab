# HBM traffic per kernel (source of bench.py's roofline.traffic): FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes
# (they do not fit one pass: microarch guide §PMC), kernel-trace only, each pass under its own timeout.
# Output: gpurun_out/pmc_fetch_write.json  {kernel: {launches, fetch_kb_avg, write_kb_avg}}  (units: KB as reported; the
# gfx950 FETCH_SIZE x2 correction is applied by the reader, not here).
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  # (--kernel-include-regex: rocprofv3 7.2 itself segfaults — host side, in a tool thread, intermittently — when it collects TCC counters on the
  #  gemm8_kernel launches of this command (measured: 0 / 5 passes complete with every kernel, 1 / 2 with the gemm8 family added to the list below,
  #  3 / 3 with the list below; SQ counters on the same kernels are fine: tools/pmc_mfma_bench.sh).  The gemm8 GEMMs' traffic is therefore not in
  #  this summary; the roofline kernel's and every other class's is.  Up to 3 attempts per pass.)
  for attempt in 1 2 3; do
    rm -rf /tmp/pmc_$c
    timeout 420 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "attention_|lm_head|gemm_bf16_glds|embed_|final_ln|sample_reduce|tok_|wordle_" --output-format csv -d /tmp/pmc_$c -- python /root/repo/bench.py --graph 0 --steps 1 --warmup 1 --no-cpu-baseline --no-train-step --no-fp32-mode --no-ppo-iteration --no-rl-reduce --no-maze > /dev/null 2>&1 && break
    echo "pass $c attempt $attempt failed/timeout"
  done
done
python - <<'PY'
import csv, glob, json, collections
out = collections.defaultdict(dict)
for c, key in (("FETCH_SIZE", "fetch_kb_avg"), ("WRITE_SIZE", "write_kb_avg")):
    fs = glob.glob(f"/tmp/pmc_{c}/**/*counter_collection.csv", recursive=True)
    if not fs:
        continue
    agg = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        if r["Counter_Name"] != c:
            continue
        k = r["Kernel_Name"][:60]
        agg[k] += float(r["Counter_Value"]); n[k] += 1
    for k in agg:
        out[k]["launches"] = n[k]; out[k][key] = agg[k] / n[k]
import sys
sys.path.insert(0, "/root/repo")
import bench
out["__meta__"] = {"csrc_digest": bench.csrc_digest(), "command": "bench.py --graph 0 --steps 1 --warmup 1 --no-cpu-baseline --no-train-step --no-fp32-mode --no-ppo-iteration --no-rl-reduce --no-maze"}
json.dump(out, open("/root/repo/gpurun_out/pmc_fetch_write.json", "w"), indent=1)
out.pop("__meta__")
for k, v in sorted(out.items(), key=lambda kv: -kv[1].get("fetch_kb_avg", 0) * kv[1].get("launches", 0))[:10]:
    print("%-60s n=%5d fetch %10.1f KB  write %10.1f KB" % (k, v.get("launches", 0), v.get("fetch_kb_avg", -1), v.get("write_kb_avg", -1)))
PY
