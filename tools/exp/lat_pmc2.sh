#!/bin/bash
# lat_pmc2.sh <n> <lib.so ...> -- LDS / wait counters of the lane-cooperative kernels at batch size n for each library (variants of tools/exp/lat_variant.sh), one
# rocprofv3 --pmc pass per counter group and library (kernel trace only beside --pmc); sums over the k_cvm* launches of the run, per library
set -eo pipefail
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
N=$1; shift
export TMPDIR=/tmp LAT_SIZES=$N
cd /tmp
for lib in "$@"; do
  tag=$(basename $lib .so)
  OUT=$GRAFT_REPO_ROOT/gpurun_out/lat_pmc2_${N}_$tag
  mkdir -p $OUT
  i=0
  for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_WAVES"; do
    i=$((i+1))
    LAT_ONLY=$GRAFT_REPO_ROOT/$lib timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT -o pmc_$i -- python3 $GRAFT_REPO_ROOT/tools/exp/lat_ab.py $GRAFT_REPO_ROOT/$lib > $OUT/pmc_$i.log 2>&1 || { echo "pmc $i FAILED for $tag"; tail -5 $OUT/pmc_$i.log; exit 1; }
  done
  python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(float); cnt = collections.Counter()
for f in sorted(glob.glob("$OUT/pmc_*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "k_cvm" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
print("== $tag, n = $N: sums over all k_cvm* launches of the run (this library alone; the same number of calls for every library)")
for k in sorted(agg):
    print(f"   {k:28s} {agg[k]:18.0f}   ({cnt[k]} launches)")
PY
done
