bash tools/collect_profiles.sh r04 cfg2 > gpurun_out/collect_cfg2.log 2>&1
bash tools/collect_profiles.sh r04 cfg3 > gpurun_out/collect_cfg3.log 2>&1
tail -3 gpurun_out/collect_cfg2.log gpurun_out/collect_cfg3.log
ls gpurun_out/profiles_r04_cfg2 gpurun_out/profiles_r04_cfg3
(timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4) > gpurun_out/full_gpu_tests.log 2>&1
cat gpurun_out/full_gpu_tests.log
