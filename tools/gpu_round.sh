#!/bin/bash
# One GPU-box round: parity tests, smoke, bench, rocprof kernel trace.  Outputs under gpurun_out/.
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 600 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 900 python bench.py --steps 5 --warmup 1 --extra > gpurun_out/bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench.log
tail -3 gpurun_out/pytest_gpu.log; tail -2 gpurun_out/smoke.log; tail -2 gpurun_out/bench.log
