#!/bin/bash
# gpurun -- bash tools/step_timeline.sh [extra bench args]: kernel trace of a short bench run -> timeline of the last optimiser step
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/timeline
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/t" -- python $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-alt "$@" > "$OUT/log.txt" 2>&1
tail -1 "$OUT/log.txt" | cut -c1-400
python $ROOT/tools/trace_step.py "$OUT/t" | tee "$OUT/step.txt"
