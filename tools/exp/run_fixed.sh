set -e
cd /root/repo
python -m pytest tests/test_gpu_fixed_g2.py tests/test_gpu_host_api.py -x -q -m gpu > gpurun_out/t_fixed_target.log 2>&1 || { tail -30 gpurun_out/t_fixed_target.log; exit 1; }
tail -3 gpurun_out/t_fixed_target.log
python tools/exp/fixed_g2_bench.py > gpurun_out/fixed_g2_bench.log 2>&1 && cat gpurun_out/fixed_g2_bench.log
