mkdir -p gpurun_out/r4m
(timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8) > gpurun_out/r4m/tests.log 2>&1
{ timeout 600 python tools/soak.py 300; timeout 600 python tools/soak.py 150 big; timeout 600 python tools/soak.py 150 big bf16; timeout 600 python tools/soak.py 200 cfg2 bf16x3; } > gpurun_out/r4m/soak.log 2>&1
cat gpurun_out/r4m/tests.log gpurun_out/r4m/soak.log
