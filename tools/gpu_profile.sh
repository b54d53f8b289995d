#!/bin/bash
# rocprofv3 evidence for bench.py (run through gpurun): kernel trace + stats, then PMC counters in separate passes
# (never --pmc together with trace domains other than --kernel-trace).  tools/summarize_prof.py turns the CSVs
# into profiles/<tag>_pmc.json + profiles/<tag>_kernel_stats.csv.
cd "$GRAFT_REPO_ROOT"
TAG=${1:-run}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
BENCH="python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra --no-power"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- $BENCH > $OUT/trace.log 2>&1; echo "trace rc=$?"
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT -o pmc_$i -- $BENCH > $OUT/pmc_$i.log 2>&1; echo "pmc $i rc=$?"
done
cd $GRAFT_REPO_ROOT
python3 tools/summarize_prof.py $OUT $TAG
