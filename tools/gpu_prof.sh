#!/bin/bash
# rocprofv3 kernel trace + stats of the bench command (summaries copied to profiles/ by hand).
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
timeout 600 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
tail -2 gpurun_out/smoke.log
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o bench -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.log 2>&1
echo "rocprof rc=$?"
cd $GRAFT_REPO_ROOT
find gpurun_out/prof -name "*.csv" | head -20
for f in $(find gpurun_out/prof -name "*kernel_stats.csv"); do echo "== $f"; head -12 $f; done
tail -2 gpurun_out/prof_bench.log | cut -c1-300
