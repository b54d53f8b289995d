#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for v in A_noinline B_inline; do
  cp tools/exp/lib$v.so plonky2-bn254-pairing_amd/libbn254_pairing_hip.so; touch plonky2-bn254-pairing_amd/libbn254_pairing_hip.so
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | grep '^{"metric' | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$v', r['value'], r['roofline']['kernel_ms_avg'])"
done; done
