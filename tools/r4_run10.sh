mkdir -p gpurun_out/r4j
timeout 1500 python tools/bf16_error_study.py > gpurun_out/r4j/bf16_study.log 2>&1
timeout 600 python tools/bf16_error_study.py cfg2 >> gpurun_out/r4j/bf16_study.log 2>&1
run() { echo "== $*" ; env "$@" AMDSPEECH_BENCH_CFG3=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   bench ms/step %.3f fwd %.3f bwd %.3f' % (d['ms_per_step'], d['config']['fwd_recurrence_ms'], d['config']['bwd_recurrence_ms']))"; }
D=$(pwd)/tools/variants/dev.so
{
run A=1
run AMDSPEECH_FLOW_DZ0=1
run AMDSPEECH_LIB=$D AMDSPEECH_FLOW_GEMM=4:34
run AMDSPEECH_LIB=$D AMDSPEECH_FLOW_GEMM=4:30
run AMDSPEECH_LIB=$D AMDSPEECH_FLOW_GEMM=4:42
run AMDSPEECH_LIB=$D AMDSPEECH_FLOW_GEMM=4:34 AMDSPEECH_FLOW_DZ0=1
run AMDSPEECH_LIB=$D AMDSPEECH_FLOW_GEMM=4:30 AMDSPEECH_FLOW_DZ0=1
run AMDSPEECH_LIB=$D AMDSPEECH_FLOW_GEMM=4:26 AMDSPEECH_FLOW_DZ0=1
run A=2
} > gpurun_out/r4j/sweep.log 2>&1
c3() { echo "== $*"; timeout 600 python bench.py "$@" --steps 4 --warmup 2 --no-cpu-baseline --no-alt 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ms/step %.2f fwd %.3f bwd %.3f' % (d['ms_per_step'], d['config']['fwd_recurrence_ms'], d['config']['bwd_recurrence_ms']))"; }
{ c3 --config cfg3; c3 --config cfg3 --precision bf16x3; c3 --config cfg3 --precision bf16; c3 --config cfg5 --precision bf16x3; c3 --config cfg5 --precision bf16; c3 --config cfg2 --precision bf16; } > gpurun_out/r4j/prec.log 2>&1
cat gpurun_out/r4j/bf16_study.log gpurun_out/r4j/sweep.log gpurun_out/r4j/prec.log
