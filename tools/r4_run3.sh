mkdir -p gpurun_out/r4c
run() { echo "== $*" ; env "$@" AMDSPEECH_BENCH_CFG3=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   bench ms/step %.3f fwd %.3f bwd %.3f' % (d['ms_per_step'], d['config']['fwd_recurrence_ms'], d['config']['bwd_recurrence_ms']))"; }
{
run AMDSPEECH_FLOW_GEMM=4:38
run AMDSPEECH_FLOW_GEMM=4:34
run AMDSPEECH_FLOW_GEMM=4:30
run AMDSPEECH_FLOW_GEMM=4:26
run AMDSPEECH_FLOW_GEMM=6:30
run AMDSPEECH_FLOW_GEMM=0:0
run AMDSPEECH_FLOW_GEMM=4:30 AMDSPEECH_FLOW_DZ0=1
run AMDSPEECH_FLOW_GEMM=4:26 AMDSPEECH_FLOW_DZ0=1
run AMDSPEECH_FLOW_GEMM=4:34
} > gpurun_out/r4c/sweep.log 2>&1
cat gpurun_out/r4c/sweep.log
