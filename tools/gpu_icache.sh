#!/bin/bash
# instruction-cache counters of the bench command (run through gpurun): bash tools/gpu_icache.sh <tag> [bench args]
cd "$GRAFT_REPO_ROOT"
TAG=${1:-run}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/icache_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra --no-power $*"
timeout 600 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQ_INSTS_VALU SQ_WAVE_CYCLES --output-format csv -d $OUT -o ic -- $BENCH > $OUT/ic.log 2>&1; echo "rc=$?"
cd $GRAFT_REPO_ROOT
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_INSTS_VALU": calls[k] += 1
for k in acc:
    print(k, "launches", calls[k])
    for c, v in sorted(acc[k].items()):
        print("   %-30s %18.0f per launch" % (c, v / max(1, calls[k])))
PY
