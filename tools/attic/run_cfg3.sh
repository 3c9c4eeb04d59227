#!/bin/bash
# gpurun -- bash tools/run_cfg3.sh name...: configs[2] step time and the per-layer recurrence kernels with each variant library
for v in "$@"; do
  lib=$(pwd)/tools/variants/$v.so
  [ "$v" = product ] && lib=$(pwd)/rnn-speech_amd/libamdspeech.so
  AMDSPEECH_LIB=$lib timeout 600 python bench.py --config cfg3 --steps 4 --warmup 1 --no-cpu-baseline --no-alt 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('== $v: cfg3 ms/step %.2f  fwd %.2f us/step  bwd %.2f us/step' % (d['ms_per_step'], d['roofline']['fwd_step']['avg_time_step_us'], d['roofline']['avg_time_step_us']))"
done
