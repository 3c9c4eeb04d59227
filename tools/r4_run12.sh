mkdir -p gpurun_out/r4l
(timeout 1800 python -m pytest tests/test_gpu_model.py tests/test_gpu_kernels.py tests/test_gpu_fullsize_cfg3.py -m gpu -x -q -k "bf16 or non_default" 2>&1 | tail -8) > gpurun_out/r4l/tests.log 2>&1
cat gpurun_out/r4l/tests.log
bash tools/collect_profiles.sh r04 cfg2 2>&1 | tail -12
bash tools/collect_profiles.sh r04 cfg3 2>&1 | tail -8
