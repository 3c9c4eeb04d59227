#!/bin/bash
# gpurun -- bash tools/run_variants.sh name1 name2 ...: recurrence-kernel times (and the bench step) with each variant library
for v in "$@"; do
  lib=$(pwd)/tools/variants/$v.so
  [ "$v" = product ] && lib=$(pwd)/rnn-speech_amd/libamdspeech.so
  if [ "$v" = old ]; then lib=$(pwd)/rnn-speech_amd/libamdspeech.so; export AMDSPEECH_FLOW_FWD_WORKERS=0; else unset AMDSPEECH_FLOW_FWD_WORKERS; fi
  echo "== $v: $(AMDSPEECH_LIB=$lib timeout 300 python tools/flow_times.py 2>&1 | tail -3 | grep -v 'per call' | tr '\n' ' ')"
  if [ -n "$BENCH" ]; then AMDSPEECH_LIB=$lib AMDSPEECH_BENCH_CFG3=0 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   bench ms/step %.3f fwd %.3f bwd %.3f loss %s' % (d['ms_per_step'], d['config']['fwd_recurrence_ms'], d['config']['bwd_recurrence_ms'], d['config'].get('mean_ctc_loss')))"; fi
done
