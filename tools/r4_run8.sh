mkdir -p gpurun_out/r4h
(timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15) > gpurun_out/r4h/tests.log 2>&1
c3() { echo "== cfg3 $*"; env "$@" timeout 600 python bench.py --config cfg3 --steps 4 --warmup 2 --no-cpu-baseline --no-alt 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   cfg3 ms/step %.2f fwd %.3f bwd %.3f' % (d['ms_per_step'], d['config']['fwd_recurrence_ms'], d['config']['bwd_recurrence_ms']))"; }
{ c3 A=1; c3 A=2; } > gpurun_out/r4h/cfg3.log 2>&1
(BENCH=1 bash tools/run_variants.sh product) > gpurun_out/r4h/variants.log 2>&1
tail -8 gpurun_out/r4h/tests.log; cat gpurun_out/r4h/variants.log gpurun_out/r4h/cfg3.log
