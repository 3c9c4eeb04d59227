mkdir -p gpurun_out/r4s
ROOT=$(pwd)
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r4s/stats5 -- python $ROOT/bench.py --config cfg5 --precision bf16 --steps 5 --warmup 2 --no-cpu-baseline --no-alt > $ROOT/gpurun_out/r4s/stats5.log 2>&1
cd $ROOT
f=$(find gpurun_out/r4s/stats5 -name "*kernel_stats.csv" | head -1)
cp $f gpurun_out/r4s/cfg5_bf16_kernel_stats.csv
head -25 gpurun_out/r4s/cfg5_bf16_kernel_stats.csv | cut -c1-200
rm -rf gpurun_out/r4s/stats5
(timeout 1200 python -m pytest tests/test_gpu_model.py -m gpu -x -q 2>&1 | tail -3) > gpurun_out/r4s/tests.log 2>&1
cat gpurun_out/r4s/tests.log
