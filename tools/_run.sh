g++ -O2 -o /tmp/beam_scale tools/beam_scale.cpp -ldl && for b in 1 32; do /tmp/beam_scale $b; done
mkdir -p gpurun_out/r4z
timeout 1500 python tools/dropin_decoder_sweep.py > gpurun_out/r4z/sweep.log 2>&1
grep -E "ms per step|rror|cpu.max" gpurun_out/r4z/sweep.log | tail -10
