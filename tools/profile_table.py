"""Per-kernel table from the committed rocprofv3 outputs: average duration, HBM-side bytes per launch
(2*FETCH_SIZE + WRITE_SIZE, gfx950 correction of MI355X_MICROARCH.md), GB/s against 8 TB/s, and for
the two recurrent step kernels the algorithmic MFMA rate against the 157.3 TFLOP/s f32 peak.
Usage: python tools/profile_table.py > profiles/r01_kernel_table.md"""
import csv
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L, H, B = 3, 512, 32
T_STEPS = 1001 + L - 1      # time steps one launch of a dataflow kernel covers (bench config)
FLOPS = {"lstm_bwd_step": (2 * L - 1) * 2.0 * B * 4 * H * H, "lstm_fwd_step": L * 2.0 * B * 2 * H * 4 * H,
         "lstm_bwd_flow": (2 * L - 1) * 2.0 * B * 4 * H * H * T_STEPS, "lstm_fwd_flow": L * 2.0 * B * 2 * H * 4 * H * T_STEPS}


def short(name):
    return name.replace("void ", "").replace("amdspeech::", "").split("(")[0]


stats = list(csv.DictReader(open(os.path.join(ROOT, "profiles", "r01_bench_kernel_stats.csv"))))
pmc = {short(k): v for k, v in json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_fetch_write_size.json"))).items()}
print("| kernel | calls | avg µs | MB / launch | HBM-side GB/s (% of 8 TB/s) | algorithmic TFLOP/s (% of 157.3) | share of step |")
print("|---|---|---|---|---|---|---|")
for r in stats:
    name = short(r["Name"])
    if "amdspeech" not in r["Name"] or name not in pmc or "FETCH_SIZE" not in pmc[name] or "WRITE_SIZE" not in pmc[name]:
        continue
    us = float(r["AverageNs"]) / 1e3
    mb = (2 * pmc[name]["FETCH_SIZE"]["mean"] + pmc[name]["WRITE_SIZE"]["mean"]) * 1024 / 1e6
    gbs = mb * 1e6 / (us * 1e-6) / 1e9
    fl = [v for k, v in FLOPS.items() if name.startswith(k)]
    tf = "%.1f (%.0f %%)" % (fl[0] / (us * 1e-6) / 1e12, fl[0] / (us * 1e-6) / 1e12 / 157.3 * 100) if fl else "—"
    print("| `%s` | %s | %.1f | %.2f | %.0f (%.0f %%) | %s | %s %% |" % (name, r["Calls"], us, mb, gbs, gbs / 80.0, tf, r["Percentage"]))
