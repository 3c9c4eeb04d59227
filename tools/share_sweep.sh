#!/bin/bash
# gpurun -- bash tools/share_sweep.sh: the in-kernel weight-gradient workers' share of the frames (development build tools/variants/dev.so,
# AMDSPEECH_DEVTRACE=9: the only builds that read AMDSPEECH_FLOW_GEMM = pieces:percent) against the bench step, alternating
for rep in 1 2; do
for g in "$@"; do
  AMDSPEECH_LIB=$(pwd)/tools/variants/dev.so AMDSPEECH_FLOW_GEMM=$g AMDSPEECH_BENCH_CFG3=0 timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-alt 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('share $g: ms/step %.3f fwd %.3f bwd %.3f' % (d['ms_per_step'], d['config']['fwd_recurrence_ms'], d['config']['bwd_recurrence_ms']))"
done
done
