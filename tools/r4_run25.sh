mkdir -p gpurun_out/r4w
OMP_NUM_THREADS=4 timeout 1200 python tools/dropin_decoder_sweep.py > gpurun_out/r4w/sweep_omp4.log 2>&1
grep -E "ms per step|rror" gpurun_out/r4w/sweep_omp4.log | tail -10
cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu.stat 2>/dev/null | head -8
