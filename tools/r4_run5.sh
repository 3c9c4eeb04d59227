mkdir -p gpurun_out/r4e
{
echo "== flow_times product"; timeout 300 python tools/flow_times.py 2>&1 | tail -2
echo "== flow_times trace4 lib"; AMDSPEECH_LIB=$(pwd)/tools/variants/trace4.so timeout 300 python tools/flow_times.py 2>&1 | tail -2
echo "== trace tool, product lib"; TRACE_DROPOUT=1 timeout 300 python tools/trace_flow2.py 2>&1 | grep "bwd kernel"
echo "== trace tool, trace4 lib"; TRACE_DROPOUT=1 AMDSPEECH_LIB=$(pwd)/tools/variants/trace4.so timeout 300 python tools/trace_flow2.py 2>&1 | grep "bwd kernel"
echo "== flow_times product, no TRACE"; AMDSPEECH_FLOW_GEMM=0:0 timeout 300 python tools/flow_times.py 2>&1 | tail -2
} > gpurun_out/r4e/cmp.log 2>&1
cat gpurun_out/r4e/cmp.log
