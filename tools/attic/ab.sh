#!/bin/bash
# usage: ab.sh label env... ; prints step/fwd/bwd
label=$1; shift
env "$@" timeout 200 python bench.py --no-alt --no-cpu-baseline --steps 12 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', round(d['ms_per_step'],3), round(d['config']['fwd_recurrence_ms'],3), round(d['config']['bwd_recurrence_ms'],3), d['config']['mean_ctc_loss'])"
