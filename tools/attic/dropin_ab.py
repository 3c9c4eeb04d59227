"""Dev tool: bench.py's extras.dropin_run_train_step alone (greedy / beam), e.g. under AMDSPEECH_FUSED_CTC=0/1."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
bench.apply_config("cfg2") if hasattr(bench, "apply_config") else None
for dec in sys.argv[1:] or ["greedy"]:
    r = bench.dropin_run_train_step(12, train_decoder=dec)
    print(dec, round(r["ms_per_step"], 3), "ms per run_train_step; host cores busy", r.get("host_cores_busy"))
