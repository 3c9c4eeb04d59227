mkdir -p gpurun_out/r4o; R=$(pwd); export TMPDIR=/tmp
pm() { # name, env...
  local name=$1; shift
  cd /tmp
  for c in WRITE_SIZE FETCH_SIZE; do
    rm -rf $R/gpurun_out/r4o/p_$c
    env "$@" timeout 200 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/r4o/p_$c -- python $R/tools/flow_times.py > $R/gpurun_out/r4o/p.log 2>&1
  done
  cd $R
  python - "$name" <<'PY'
import csv, glob, sys, statistics
out={}
for c in ('WRITE_SIZE','FETCH_SIZE'):
    for f in glob.glob('gpurun_out/r4o/p_%s/**/*counter_collection.csv'%c, recursive=True):
        for r in csv.DictReader(open(f)):
            if 'lstm_bwd_flow2' in r['Kernel_Name']: out.setdefault(c,[]).append(float(r['Counter_Value']))
w=statistics.median(out.get('WRITE_SIZE',[0])); f=statistics.median(out.get('FETCH_SIZE',[0]))
print('%-12s bwd_flow2: WRITE %.2f GB  FETCHx2 %.2f GB  -> %.2f MB per time step' % (sys.argv[1], w*1024/1e9, 2*f*1024/1e9, (2*f+w)*1024/1003/1e6))
PY
  rm -rf gpurun_out/r4o/p_WRITE_SIZE gpurun_out/r4o/p_FETCH_SIZE
}
{
pm product A=1
pm stash-nt AMDSPEECH_LIB=$R/tools/variants/stnt.so
pm stash+dg-nt AMDSPEECH_LIB=$R/tools/variants/stnt2.so
pm no-workers AMDSPEECH_LIB=$R/tools/variants/dev.so AMDSPEECH_FLOW_GEMM=0:0
BENCH=1 bash tools/run_variants.sh product stnt stnt2
} > gpurun_out/r4o/traffic.log 2>&1
cat gpurun_out/r4o/traffic.log
