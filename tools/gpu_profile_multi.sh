#!/bin/bash
# PMC counters of the Groth16-shape kernel (k_mpairing, 2^18 groups x 4 pairs: bench.py's configs[3] extra), run through gpurun.
cd "$GRAFT_REPO_ROOT"
TAG=${1:-multi}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-power --no-scalar-latency --no-host-path"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- $BENCH > $OUT/trace.log 2>&1; echo "trace rc=$?"
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT -o pmc_$i -- $BENCH > $OUT/pmc_$i.log 2>&1; echo "pmc $i rc=$?"
done
cd $GRAFT_REPO_ROOT
PROF_KERNEL="::k_mpairing(" PROF_LOG2_BATCH=18 PROF_K=4 python3 tools/summarize_prof.py $OUT $TAG
