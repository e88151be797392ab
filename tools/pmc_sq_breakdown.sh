cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pmc1 -- python /root/repo/bench.py --graph 0 --steps 1 --warmup 1 --no-cpu-baseline > /root/repo/gpurun_out/pmc1.log 2>&1
f=$(find /tmp/pmc1 -name "*counter_collection.csv" | head -1); ls -la $f
python - $f <<'PY'
import csv,sys,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
rows=list(csv.DictReader(open(sys.argv[1])))
print(rows[0].keys())
for r in rows:
    k=r["Kernel_Name"].split("(")[0][-58:]
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"])
    if r["Counter_Name"]=="SQ_WAVES": cnt[k]+=1
for k,n in cnt.most_common(14):
    a=agg[k]; wc=a["SQ_WAVE_CYCLES"] or 1
    print("%-58s n=%5d waves/launch=%7.0f wait_any=%4.1f%% wait_inst=%4.1f%% active=%4.1f%% gui_active/launch=%8.0f busy/launch=%8.0f" % (k,n,a["SQ_WAVES"]/n,100*a["SQ_WAIT_ANY"]/wc,100*a["SQ_WAIT_INST_ANY"]/wc,100*a["SQ_ACTIVE_INST_ANY"]/wc,a["GRBM_GUI_ACTIVE"]/n,a["SQ_BUSY_CYCLES"]/n))
PY
