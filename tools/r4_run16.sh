mkdir -p gpurun_out/r4p
(timeout 1200 python -m pytest tests/test_gpu_model.py -m gpu -x -q -k "adam_parity" 2>&1 | tail -12) > gpurun_out/r4p/tests.log 2>&1
cat gpurun_out/r4p/tests.log
