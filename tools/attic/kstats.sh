#!/bin/bash
# dev: per-kernel average durations of `python tools/quick_bench.py "$@"` (rocprofv3 kernel trace)
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/ks
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python $GRAFT_REPO_ROOT/tools/quick_bench.py "$@" > /tmp/ks.log 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('/tmp/ks/**/*kernel_stats.csv',recursive=True)
for r in list(csv.DictReader(open(f[0])))[:8]:
    print("%-60s calls %6s avg us %10.1f  %5s%%" % (r['Name'].replace('void amdspeech::','')[:60], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
PY
