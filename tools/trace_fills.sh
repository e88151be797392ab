#!/bin/bash
# which fills / copies does one ILQL bf16 step issue?  kernel-trace rows of FillFunctor / copyBuffer kernels with their grid sizes -> stdout
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trf
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/trf -- python $REPO/bench.py --mode ilql-step --train-matmul bf16 --steps 1 --warmup 1 > /dev/null 2>&1
python - $(find /tmp/trf -name "*kernel_trace.csv" | head -1) <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
half = len(rows) // 2
agg = collections.Counter(); tim = collections.Counter()
for r in rows[half:]:
    n = r["Kernel_Name"]
    if "Fill" in n or "copyBuffer" in n or "elementwise" in n:
        key = (n[:60], int(r["Grid_Size_X"]) * int(r.get("Workgroup_Size_X", 1) or 1) if False else int(r["Grid_Size_X"]))
        agg[key] += 1; tim[key] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, c in sorted(agg.items(), key=lambda kv: -tim[kv[0]])[:30]:
    print("%-62s grid %10d  n=%4d  total %8.1f us" % (k[0], k[1], c, tim[k] / 1e3))
PY
