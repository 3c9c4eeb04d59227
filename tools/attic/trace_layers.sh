# gpurun -- bash tools/trace_layers.sh <dev5 variant>: per-phase stamps of every layer's traced workgroup and worker
for l in 0 1 2; do echo "=== layer $l"; AMDSPEECH_TRACE_LAYER=$l AMDSPEECH_LIB=$(pwd)/tools/variants/$1.so timeout 200 python tools/trace_fwd2.py 2>&1 | grep -v amdgpu.ids | tail -33; done
