mkdir -p gpurun_out/r4w
timeout 1500 python tools/dropin_decoder_sweep.py > gpurun_out/r4w/sweep.log 2>&1
grep -E "ms per step|rror|cpu.max" gpurun_out/r4w/sweep.log | tail -10
