#!/bin/bash
# Regenerates the evidence under profiles/ on a GPU box (run from the repo root through gpurun):
#   tools/collect_profiles.sh r01
# Three separate rocprofv3 passes of the SAME bench command (kernel stats; FETCH_SIZE; WRITE_SIZE --
# counters never combined with other trace domains), plus one un-profiled bench line.
# Outputs land in gpurun_out/profiles_<tag>/ ; copy them into profiles/ afterwards.
set -u
TAG=${1:-r01}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-alt"
cd /tmp      # rocprofv3 counter passes crash from other working directories on this image

timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $CMD > "$OUT/stats.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- $CMD --sync-each-step > "$OUT/fetch.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- $CMD --sync-each-step > "$OUT/write.log" 2>&1
timeout 900 python $ROOT/bench.py > "$OUT/bench_line.json" 2> "$OUT/bench.log"

python - "$OUT" "$TAG" <<'EOF'
import csv, glob, json, os, shutil, statistics, sys
out, tag = sys.argv[1], sys.argv[2]
st = glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True)
if st:
    shutil.copy(st[0], os.path.join(out, "%s_bench_kernel_stats.csv" % tag))
pmc = {}
for leg in ("fetch", "write"):
    for f in glob.glob(os.path.join(out, leg, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].replace("void ", "").split("(")[0]
            if "amdspeech" not in name:
                continue
            pmc.setdefault(name, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
red = {k: {c: {"calls": len(v), "median": statistics.median(v), "mean": sum(v) / len(v)} for c, v in d.items()}
       for k, d in pmc.items()}
json.dump(red, open(os.path.join(out, "%s_pmc_fetch_write_size.json" % tag), "w"), indent=1, sort_keys=True)
for leg in ("stats", "fetch", "write"):          # raw traces are large; keep the reductions only
    shutil.rmtree(os.path.join(out, leg), ignore_errors=True)
print(open(os.path.join(out, "bench_line.json")).read()[:1500])
EOF
