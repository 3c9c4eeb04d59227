mkdir -p gpurun_out/r4r
(BENCH=1 bash tools/run_variants.sh product fold pre1 pre2 product fold pre1 pre2) > gpurun_out/r4r/var.log 2>&1
cat gpurun_out/r4r/var.log
for v in pre1 pre2; do
(AMDSPEECH_LIB=$(pwd)/tools/variants/$v.so timeout 1200 python -m pytest tests/test_gpu_model.py -m gpu -x -q -k "adam_parity or dropout or bidirectional" 2>&1 | tail -3) > gpurun_out/r4r/tests_$v.log 2>&1
cat gpurun_out/r4r/tests_$v.log
done
