set -u
export TMPDIR=/tmp
ROOT=$(pwd)
mkdir -p gpurun_out/p5
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_suite_final.txt 2>&1; grep -E "passed|failed" gpurun_out/gpu_suite_final.txt | tail -1
bash tools/collect_profiles.sh r05 cfg2 > gpurun_out/collect_cfg2.log 2>&1; tail -1 gpurun_out/collect_cfg2.log | cut -c1-260
bash tools/collect_profiles.sh r05 cfg3 > gpurun_out/collect_cfg3.log 2>&1; tail -1 gpurun_out/collect_cfg3.log | cut -c1-260
for cfg in cfg5 cfg3; do
  cd /tmp
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/p5/stats_$cfg -- python $ROOT/bench.py --config $cfg --precision bf16 --steps 6 --warmup 2 --no-cpu-baseline --no-alt > $ROOT/gpurun_out/p5/log_$cfg.txt 2>&1
  cd $ROOT
  f=$(find gpurun_out/p5/stats_$cfg -name "*kernel_stats.csv" | head -1)
  cp $f gpurun_out/p5/r05_${cfg}_bf16_kernel_stats.csv
  rm -rf gpurun_out/p5/stats_$cfg
  grep '^{' gpurun_out/p5/log_$cfg.txt | tail -1 | cut -c1-260
done
