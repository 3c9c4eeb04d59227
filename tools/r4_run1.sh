mkdir -p gpurun_out/r4a
(timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_dropout_oracle.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -25) > gpurun_out/r4a/tests.log 2>&1
(BENCH=1 bash tools/run_variants.sh product check) > gpurun_out/r4a/variants.log 2>&1
tail -5 gpurun_out/r4a/tests.log; cat gpurun_out/r4a/variants.log
