#!/bin/bash
# pmc_variant.sh <tag> <lib.so> [AB_K] -- rocprofv3 counters of one library variant under tools/exp/ab_bench.py (run on the GPU box).
# Separate --pmc passes (kernel trace only).  Prints per-launch averages of the dominant kernel.
set -eo pipefail
: "${GRAFT_REPO_ROOT:?pmc_variant.sh runs on the GPU box through gpurun (GRAFT_REPO_ROOT is not set)}"
cd "$GRAFT_REPO_ROOT"
[ -f "$2" ] || { echo "pmc_variant.sh: library $2 not found" >&2; exit 1; }
TAG=$1; LIB=$(realpath $2); K=${3:-4}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcv_$TAG
mkdir -p $OUT
export TMPDIR=/tmp AB_NOCHECK=1 AB_K=$K AB_LOG2=20 AB_REPS=2
cd /tmp
i=0
for grp in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_ANY" "SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_TC_INST_REQ SQC_TC_STALL SQC_ICACHE_BUSY_CYCLES SQC_ICACHE_MISSES_DUPLICATE" "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_BUSY_avr" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT -o pmc_$i -- python3 $GRAFT_REPO_ROOT/tools/exp/ab_bench.py $LIB > $OUT/pmc_$i.log 2>&1 && echo "pmc $i ok" || { echo "pmc $i FAILED"; tail -5 $OUT/pmc_$i.log; exit 1; }
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in sorted(glob.glob("$OUT/pmc_*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "k_mpairing" in r["Kernel_Name"] or ("$K" == "1" and "k_pairing" in r["Kernel_Name"]):
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("== $TAG")
for k, v in sorted(agg.items()):
    print(f"{k:36s} {sum(v)/len(v):18.1f}  (n={len(v)})")
PY
