timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/full_gpu_tests.log 2>&1
grep -n "passed\|failed\|rror" gpurun_out/full_gpu_tests.log | tail -5
