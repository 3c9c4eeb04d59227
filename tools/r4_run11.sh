mkdir -p gpurun_out/r4k
(timeout 1800 python -m pytest tests/test_gpu_model.py tests/test_gpu_kernels.py tests/test_gpu_fullsize_cfg3.py -m gpu -x -q -k "bf16" 2>&1 | tail -12) > gpurun_out/r4k/tests.log 2>&1
c3() { echo "== $*"; env "$@" timeout 600 python bench.py --config cfg3 --steps 4 --warmup 2 --no-cpu-baseline --no-alt 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ms/step %.2f fwd %.3f bwd %.3f' % (d['ms_per_step'], d['config']['fwd_recurrence_ms'], d['config']['bwd_recurrence_ms']))"; }
{ c3 A=1; c3 AMDSPEECH_LIB=$(pwd)/tools/variants/near2.so; c3 A=2; c3 AMDSPEECH_LIB=$(pwd)/tools/variants/near2.so; } > gpurun_out/r4k/near2.log 2>&1
AMDSPEECH_LIB=$(pwd)/tools/variants/near2.so timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_dropout_oracle.py -m gpu -x -q -k "1024" 2>&1 | tail -3 >> gpurun_out/r4k/near2.log
cat gpurun_out/r4k/tests.log gpurun_out/r4k/near2.log
