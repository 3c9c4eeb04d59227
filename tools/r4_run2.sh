mkdir -p gpurun_out/r4b
(BENCH=1 bash tools/run_variants.sh product dl3) > gpurun_out/r4b/variants.log 2>&1
for v in trace4 trace3; do
  echo "== $v" >> gpurun_out/r4b/trace.log
  AMDSPEECH_LIB=$(pwd)/tools/variants/$v.so timeout 300 python tools/trace_flow2.py >> gpurun_out/r4b/trace.log 2>&1
done
cat gpurun_out/r4b/variants.log; cat gpurun_out/r4b/trace.log
