mkdir -p gpurun_out/r4v
nproc > gpurun_out/r4v/scale.log; lscpu | grep -E "Model name|Thread|Core|Socket|MHz" >> gpurun_out/r4v/scale.log
g++ -O2 -o /tmp/beam_scale tools/beam_scale.cpp -ldl && for b in 1 2 4 8 16 32 64; do /tmp/beam_scale $b; done >> gpurun_out/r4v/scale.log 2>&1
cat gpurun_out/r4v/scale.log
