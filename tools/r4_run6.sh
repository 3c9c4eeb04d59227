mkdir -p gpurun_out/r4f
(timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_dropout_oracle.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -5) > gpurun_out/r4f/tests.log 2>&1
(BENCH=1 bash tools/run_variants.sh product wo0 dl3 check) > gpurun_out/r4f/variants.log 2>&1
for l in 2 1; do AMDSPEECH_TRACE_LAYER=$l TRACE_DROPOUT=1 AMDSPEECH_LIB=$(pwd)/tools/variants/trace4.so timeout 300 python tools/trace_flow2.py 2>&1 | grep -v amdgpu.ids; done > gpurun_out/r4f/trace.log 2>&1
tail -3 gpurun_out/r4f/tests.log; cat gpurun_out/r4f/variants.log; cat gpurun_out/r4f/trace.log
