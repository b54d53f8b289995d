#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/bench_only.log 2>&1; tail -1 gpurun_out/bench_only.log | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('pairings/s', r['value'], 'kernel ms', r['roofline']['kernel_ms_avg'])"
