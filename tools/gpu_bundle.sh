#!/bin/bash
# gpu_bundle.sh <tag> -- ONE-LEASE evidence bundle (run through gpurun, one call = one box):
#   calibration (tools/valu_calib --mad-only) -> rocprofv3 kernel trace + PMC passes of k_pairing and of k_mpairing (separate
#   passes, --kernel-trace only beside --pmc) -> bench.py --steps 20 (reading those fresh summaries) -> in-kernel clock stamps (diagnostic library, if built) ->
#   calibration again -> tools/make_bundle.py: gpurun_out/bundle_<tag>/<tag>_bundle.json with the calibrated peak, the bench line,
#   the profiled kernel times, the counters, the in-kernel clock and the box id.  Copy the *_bundle.json, *_pmc.json and
#   *_kernel_stats.csv into profiles/ afterwards.  Every step is joined with && : a failed or killed GPU step ends the call.
set -o pipefail
cd "$GRAFT_REPO_ROOT"
TAG=${1:-r05}
OUT=$GRAFT_REPO_ROOT/gpurun_out/bundle_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" &&
timeout -k 10 120 tools/valu_calib --mad-only --json > $OUT/calib_start.json && cat $OUT/calib_start.json &&
bash tools/gpu_profile.sh ${TAG} > $OUT/profile.log 2>&1 && tail -3 $OUT/profile.log &&
bash tools/gpu_profile_multi.sh ${TAG}_groth16 > $OUT/profile_groth16.log 2>&1 && tail -3 $OUT/profile_groth16.log &&
cp gpurun_out/prof_${TAG}/${TAG}_pmc.json gpurun_out/prof_${TAG}/${TAG}_kernel_stats.csv $OUT/ &&
cp gpurun_out/prof_${TAG}_groth16/${TAG}_groth16_pmc.json gpurun_out/prof_${TAG}_groth16/${TAG}_groth16_kernel_stats.csv $OUT/ &&
BENCH_PMC_SUMMARY=$OUT/${TAG}_pmc.json BENCH_PMC_SUMMARY_GROTH16=$OUT/${TAG}_groth16_pmc.json timeout -k 10 900 python bench.py --steps 20 --warmup 2 > $OUT/bench.log 2> $OUT/bench.err &&
grep '^{"metric' $OUT/bench.log > $OUT/bench.json && echo "bench done" &&
{ if [ -f build/variants/lib_stamp.so ]; then timeout -k 10 300 python tools/clock_stamp.py build/variants/lib_stamp.so > $OUT/clock_stamp.json 2> $OUT/clock_stamp.err; else echo '{}' > $OUT/clock_stamp.json; fi; } &&
timeout -k 10 120 tools/valu_calib --mad-only --json > $OUT/calib_end.json &&
python tools/make_bundle.py $OUT $TAG
