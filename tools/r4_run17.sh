mkdir -p gpurun_out/r4q
(BENCH=1 bash tools/run_variants.sh product pre product pre) > gpurun_out/r4q/var.log 2>&1
cat gpurun_out/r4q/var.log
(AMDSPEECH_LIB=$(pwd)/tools/variants/pre.so timeout 1200 python -m pytest tests/test_gpu_model.py -m gpu -x -q -k "adam_parity or dropout or bidirectional" 2>&1 | tail -12) > gpurun_out/r4q/tests.log 2>&1
cat gpurun_out/r4q/tests.log
(AMDSPEECH_LIB=$(pwd)/tools/variants/pre.so timeout 600 python tools/soak.py 100 2>&1 | tail -3) > gpurun_out/r4q/soak.log 2>&1
cat gpurun_out/r4q/soak.log
