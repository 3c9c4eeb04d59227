#!/bin/bash
# Regenerates the evidence under profiles/ on a GPU box (run from the repo root through gpurun):
#   tools/collect_profiles.sh r02 [cfg2|cfg3]
# Separate rocprofv3 passes of the SAME bench command (kernel stats; then one --pmc pass per counter group --
# counters are never combined with other trace domains), then one un-profiled bench line (with this run's counters in profiles/).
# Outputs land in gpurun_out/profiles_<tag>/ ; copy them into profiles/ afterwards.
set -u
TAG=${1:-r02}
CFG=${2:-cfg2}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/profiles_${TAG}_$CFG
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="python $ROOT/bench.py --config $CFG --steps 10 --warmup 2 --no-cpu-baseline --no-alt"
cd /tmp      # rocprofv3 counter passes crash from other working directories on this image

timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $CMD > "$OUT/stats.log" 2>&1
pass() {   # name, counters...
    local name=$1; shift
    timeout 900 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -- $CMD --sync-each-step > "$OUT/$name.log" 2>&1
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE
pass l2 TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum

cd "$ROOT"
python - "$OUT" "$TAG" "$CFG" <<'EOF'
import csv, glob, json, os, shutil, statistics, sys
out, tag, cfg = sys.argv[1], sys.argv[2], sys.argv[3]
st = glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True)
if st:
    shutil.copy(st[0], os.path.join(out, "%s_%s_kernel_stats.csv" % (tag, cfg)))
pmc = {}
for leg in ("fetch", "write", "mfma", "l2"):
    for f in glob.glob(os.path.join(out, leg, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].replace("void ", "").split("(")[0]
            if "amdspeech" not in name:
                continue
            pmc.setdefault(name, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
red = {k: {c: {"calls": len(v), "median": statistics.median(v), "mean": sum(v) / len(v)} for c, v in d.items()}
       for k, d in pmc.items()}
for k, d in red.items():
    # measured MFMA-pipe utilisation of the kernel: cycles a CU's matrix pipes were busy / cycles the CUs were busy.
    # SQ_VALU_MFMA_BUSY_CYCLES sums the 4 SIMDs of a CU (guide: "counts cycles"), SQ_BUSY_CU_CYCLES counts per CU.
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "SQ_BUSY_CU_CYCLES" in d and d["SQ_BUSY_CU_CYCLES"]["median"] > 0:
        d["mfma_util"] = d["SQ_VALU_MFMA_BUSY_CYCLES"]["median"] / (4.0 * d["SQ_BUSY_CU_CYCLES"]["median"])
# stamp: the kernel sources these counters were collected on (bench.py prints it beside the figures it reads from here)
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
try:
    import bench as _bench
    red["_meta"] = {"kernel_src_sha16": _bench.kernel_src_sha16(), "command": "bench.py --config %s --steps 10 --warmup 2 --no-cpu-baseline --no-alt" % cfg}
except Exception as exc:
    red["_meta"] = {"kernel_src_sha16": None, "error": repr(exc)[:200]}
json.dump(red, open(os.path.join(out, "%s_pmc_%s.json" % (tag, cfg)), "w"), indent=1, sort_keys=True)
for leg in ("stats", "fetch", "write", "mfma", "l2"):          # raw traces are large; keep the reductions only
    shutil.rmtree(os.path.join(out, leg), ignore_errors=True)
# the un-profiled bench line comes LAST, with this run's counters in place: its roofline.traffic is then from the same build and box
shutil.copy(os.path.join(out, "%s_pmc_%s.json" % (tag, cfg)), os.path.join("profiles", "%s_pmc_%s.json" % (tag, cfg)))
for k, d in sorted(red.items()):
    if "lstm" in k and isinstance(d, dict):
        print(k, {c: (v if isinstance(v, float) else v["median"]) for c, v in d.items()})
EOF
timeout 900 python $ROOT/bench.py --config $CFG 2> "$OUT/bench.log" | grep '^{' | tail -1 > "$OUT/bench_line.json"
cut -c1-600 "$OUT/bench_line.json"
