#!/bin/bash
# Full GPU round (run through gpurun): parity tests, smoke, bench (2^20 headline + extra configs), rocprofv3 profile.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
TAG=${1:-r03}
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py --steps 5 --warmup 1 > gpurun_out/bench_$TAG.log 2>&1; echo "bench rc=$?"; grep '^{"metric' gpurun_out/bench_$TAG.log > gpurun_out/bench_$TAG.json; python - <<PY
import json
r=json.load(open("gpurun_out/bench_$TAG.json"))
print("pairings/s", r["value"], "ms/step", r["ms_per_step"], "roofline", r["roofline"]["frac"], "verified", r.get("verified_vs_oracle"))
print(json.dumps(r.get("cpu_baseline"), indent=1))
print(json.dumps(r.get("extra"), indent=1))
PY
tail -3 gpurun_out/bench_$TAG.log | cut -c1-300
bash tools/gpu_profile.sh $TAG 2>&1 | tail -24
