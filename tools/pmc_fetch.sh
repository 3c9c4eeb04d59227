#!/bin/bash
# gpurun -- bash tools/pmc_fetch.sh <label> [env...]: FETCH_SIZE / WRITE_SIZE of the two recurrence launches for one bench configuration
label=$1; shift
export TMPDIR=/tmp
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/pmcq_$label
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  env "$@" AMDSPEECH_BENCH_CFG3=0 timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/$c" -- python $ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-alt --sync-each-step > "$OUT/$c.log" 2>&1
done
cd "$ROOT"
python - "$OUT" "$label" <<'PY'
import csv, glob, os, statistics, sys
out, label = sys.argv[1], sys.argv[2]
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(os.path.join(out, c, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            if "lstm_bwd_flow2" in n or "lstm_fwd_flow2" in n or "tn_group" in n:
                res.setdefault((n.split("(")[0][-40:], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
for k, v in sorted(res.items()):
    print(label, k[0], k[1], "median %.3f GB (x2 for FETCH)" % (statistics.median(v) / 1e6 * 1.024), "calls", len(v))
PY
rm -rf "$OUT"/FETCH_SIZE "$OUT"/WRITE_SIZE
