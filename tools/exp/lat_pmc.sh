#!/bin/bash
# lat_pmc.sh [n] -- rocprofv3 counters of the lane-cooperative kernel (k_cvm) at batch size n (default 1), run on the GPU box.
set -eo pipefail
: "${GRAFT_REPO_ROOT:?lat_pmc.sh runs on the GPU box through gpurun}"
N=${1:-1}
OUT=$GRAFT_REPO_ROOT/gpurun_out/lat_pmc_$N
mkdir -p $OUT
export TMPDIR=/tmp LAT_SIZES=$N
cd /tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_LDS_UNALIGNED_STALL" "SQ_IFETCH SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_INSTS_SMEM SQ_WAVES"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT -o pmc_$i -- python3 $GRAFT_REPO_ROOT/tools/exp/lat_ab.py > $OUT/pmc_$i.log 2>&1 && echo "pmc $i ok" || { echo "pmc $i FAILED"; tail -5 $OUT/pmc_$i.log; exit 1; }
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in sorted(glob.glob("$OUT/pmc_*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "k_cvm" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("== k_cvm (pairing program) n=$N")
for k, v in sorted(agg.items()):
    print(f"{k:36s} {sum(v)/len(v):18.1f}  (n={len(v)})")
PY
