#!/bin/bash
# rocprofv3 evidence for bench.py: kernel trace + stats, then PMC counters in separate passes
# (no --pmc together with trace domains other than --kernel-trace).  Copy summaries to profiles/.
cd "$GRAFT_REPO_ROOT"
TAG=${1:-run}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- $BENCH > $OUT/trace.log 2>&1; echo "trace rc=$?"
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_INST_CYCLES_SALU"; do
  name=$(echo $grp | cut -d' ' -f1)
  timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT -o pmc_$name -- $BENCH > $OUT/pmc_$name.log 2>&1; echo "pmc $name rc=$?"
done
cd $GRAFT_REPO_ROOT
ls $OUT | head -40
python3 - <<'PY'
import csv, glob, os, collections
out=os.environ.get("OUT") or sorted(glob.glob("gpurun_out/prof_*"))[-1]
for f in sorted(glob.glob(out+"/pmc_*counter_collection.csv")):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k2_pairing" in r.get("Kernel_Name","") or "k_pairing" in r.get("Kernel_Name",""):
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in agg.items(): print(os.path.basename(f), k, "per-launch avg", sum(v)/len(v), "n", len(v))
PY
for f in $(find $OUT -name "*kernel_stats.csv"); do head -4 $f | cut -c1-300; done
