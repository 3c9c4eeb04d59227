mkdir -p gpurun_out/r4u
(timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_kernels.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3) > gpurun_out/r4u/tests.log 2>&1
cat gpurun_out/r4u/tests.log
for cfg in "cfg5 bf16" "cfg3 bf16" "cfg5 bf16x3"; do set -- $cfg; echo "== $cfg"; timeout 600 python bench.py --config $1 --precision $2 --steps 4 --warmup 2 --no-cpu-baseline --no-alt 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   ms/step %.2f' % d['ms_per_step'])"; done > gpurun_out/r4u/bf.log 2>&1
cat gpurun_out/r4u/bf.log
