#!/bin/bash
cd "$GRAFT_REPO_ROOT"; OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_icache; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
BENCH="python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_IFETCH --output-format csv -d $OUT -o ic -- $BENCH > $OUT/ic.log 2>&1; echo rc=$?
tail -3 $OUT/ic.log | cut -c1-200
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv, glob, collections
agg=collections.defaultdict(list)
for f in glob.glob("gpurun_out/prof_icache/ic*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k3_pairing" in r.get("Kernel_Name",""): agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items(): print(k, sum(v)/len(v))
PY
