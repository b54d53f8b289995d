#!/bin/bash
# Samples power / clocks (rocm-smi) while bench.py runs its 2^20-pairing steps (run through gpurun): is the kernel power-limited?
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
(python bench.py --steps 80 --warmup 2 --no-extra --no-cpu-baseline > gpurun_out/power_bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/power_bench.log) &
BP=$!
T0=$(date +%s.%N)
while kill -0 $BP 2>/dev/null; do
  T=$(date +%s.%N)
  echo -n "t=$(echo "$T - $T0" | bc) "
  rocm-smi --showpower --showclocks 2>&1 | grep -E "Power \(W\)|sclk" | sed 's/GPU\[0\]\s*: //' | tr '\n' ';'; echo
done > gpurun_out/power_samples.txt
wait $BP
rocm-smi --showmaxpower 2>&1 | grep -i "max"
awk 'NR%3==0' gpurun_out/power_samples.txt | head -60
grep -o '"ms_per_step": [0-9.]*' gpurun_out/power_bench.log
