mkdir -p gpurun_out/r4i
(timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > gpurun_out/r4i/tests.log 2>&1
(timeout 600 python -m pytest tests/test_gpu_dropout_oracle.py -m gpu -q -s -k headline 2>&1 | grep -i "dlogits rel\|passed\|failed") > gpurun_out/r4i/ctc_err.log 2>&1
echo "== gemm product" > gpurun_out/r4i/gemm.log; timeout 300 python tools/gemm_f32_bench.py >> gpurun_out/r4i/gemm.log 2>&1
echo "== gemm dev KC_MIN_K=1024" >> gpurun_out/r4i/gemm.log; AMDSPEECH_LIB=$(pwd)/tools/variants/dev.so AMDSPEECH_KC_MIN_K=1024 timeout 300 python tools/gemm_f32_bench.py >> gpurun_out/r4i/gemm.log 2>&1
python tools/ctc_time.py > gpurun_out/r4i/ctc_time.log 2>&1
tail -8 gpurun_out/r4i/tests.log; cat gpurun_out/r4i/ctc_err.log gpurun_out/r4i/gemm.log gpurun_out/r4i/ctc_time.log
