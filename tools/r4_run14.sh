mkdir -p gpurun_out/r4n; R=$(pwd); cd /tmp; export TMPDIR=/tmp
for c in WRITE_SIZE FETCH_SIZE; do
  rm -rf $R/gpurun_out/r4n/wb_$c
  timeout 120 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/r4n/wb_$c -- $R/tools/l2_writeback_probe > $R/gpurun_out/r4n/wb_$c.log 2>&1
done
cd $R; python - <<'PY'
import csv, glob
for c in ('WRITE_SIZE','FETCH_SIZE'):
    for f in glob.glob('gpurun_out/r4n/wb_%s/**/*counter_collection.csv'%c, recursive=True):
        for r in csv.DictReader(open(f)):
            print(r['Kernel_Name'][:40], r['Counter_Name'], r['Counter_Value'], 'KB')
PY
tail -1 gpurun_out/r4n/wb_WRITE_SIZE.log
rm -rf gpurun_out/r4n/wb_WRITE_SIZE gpurun_out/r4n/wb_FETCH_SIZE
