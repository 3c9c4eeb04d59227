mkdir -p gpurun_out/r4d
{
for l in 2 1 0; do
  AMDSPEECH_TRACE_LAYER=$l TRACE_DROPOUT=1 AMDSPEECH_LIB=$(pwd)/tools/variants/trace4.so timeout 300 python tools/trace_flow2.py 2>&1 | grep -v amdgpu.ids
done
AMDSPEECH_TRACE_LAYER=1 AMDSPEECH_LIB=$(pwd)/tools/variants/trace4.so timeout 300 python tools/trace_flow2.py 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r4d/trace.log 2>&1
cat gpurun_out/r4d/trace.log
