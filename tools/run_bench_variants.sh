#!/bin/bash
# gpurun -- bash tools/run_bench_variants.sh [cfg] name1 name2 ...: the bench step with each variant library (tools/variants/<name>.so; "product" = the in-tree one)
cfg=cfg2; case "$1" in cfg2|cfg3|cfg5) cfg=$1; shift;; esac
for v in "$@"; do
  lib=$(pwd)/tools/variants/$v.so
  [ "$v" = product ] && lib=$(pwd)/rnn-speech_amd/libamdspeech.so
  AMDSPEECH_LIB=$lib AMDSPEECH_BENCH_CFG3=0 timeout 600 python bench.py --config $cfg --steps ${STEPS:-12} --warmup 3 --no-cpu-baseline --no-alt $EXTRA 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v: ms/step %.3f fwd %.3f bwd %.3f loss %s' % (d['ms_per_step'], d['config']['fwd_recurrence_ms'], d['config']['bwd_recurrence_ms'], d['config'].get('mean_ctc_loss')))"
done
