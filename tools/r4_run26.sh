mkdir -p gpurun_out/r4x
(timeout 900 python -m pytest tests/test_gpu_dropin.py tests/test_gpu_bench.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3) > gpurun_out/r4x/tests.log 2>&1
cat gpurun_out/r4x/tests.log
python - <<'PY' > gpurun_out/r4x/sweep.log 2>&1
import os, sys
sys.path.insert(0, os.getcwd())
import bench
print("greedy %.2f" % bench.dropin_run_train_step(20, "greedy")["ms_per_step"], flush=True)
for lag in (1, 2):
    os.environ["AMDSPEECH_TRAIN_DECODER_LAG"] = str(lag)
    print("beam lag %d: %.2f" % (lag, bench.dropin_run_train_step(20, "beam")["ms_per_step"]), flush=True)
PY
grep -E "greedy|beam" gpurun_out/r4x/sweep.log
g++ -O2 -o /tmp/beam_scale tools/beam_scale.cpp -ldl && for b in 1 32; do /tmp/beam_scale $b; done
