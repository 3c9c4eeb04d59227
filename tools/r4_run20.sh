mkdir -p gpurun_out/r4t
c3() { echo "== $*"; env "$@" timeout 600 python bench.py --config cfg3 --steps 4 --warmup 2 --no-cpu-baseline --no-alt 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ms/step %.2f fwd %.3f bwd %.3f' % (d['ms_per_step'], d['config']['fwd_recurrence_ms'], d['config']['bwd_recurrence_ms']))"; }
c2() { echo "== $*"; env "$@" AMDSPEECH_BENCH_CFG3=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   bench ms/step %.3f fwd %.3f bwd %.3f' % (d['ms_per_step'], d['config']['fwd_recurrence_ms'], d['config']['bwd_recurrence_ms']))"; }
{ c3 A=1; c3 AMDSPEECH_LIB=$(pwd)/tools/variants/nosettle.so; c3 A=2; c3 AMDSPEECH_LIB=$(pwd)/tools/variants/nosettle.so;
  rocm-smi --showclocks 2>&1 | grep -i "sclk\|mclk\|fclk" | head -8; rocm-smi --showperflevel 2>&1 | grep -i perf | head -3
  c2 A=1; 
  rocm-smi --setperflevel high 2>&1 | tail -2; rocm-smi --showperflevel 2>&1 | grep -i perf | head -3
  c2 A=2; c3 A=3;
  rocm-smi --setperflevel auto 2>&1 | tail -1; } > gpurun_out/r4t/ab.log 2>&1
cat gpurun_out/r4t/ab.log
(timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_dropout_oracle.py tests/test_gpu_fullsize_cfg3.py -m gpu -x -q 2>&1 | tail -3) > gpurun_out/r4t/tests.log 2>&1
cat gpurun_out/r4t/tests.log
(timeout 600 python tools/soak.py 100 big 2>&1 | tail -1; timeout 600 python tools/soak.py 200 2>&1 | tail -1) > gpurun_out/r4t/soak.log 2>&1
cat gpurun_out/r4t/soak.log
