cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/ks
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d /tmp/ks -- python $GRAFT_REPO_ROOT/tools/quick_bench.py --stream > /tmp/ks.log 2>&1
python - <<'PY'
import csv,glob
f=glob.glob('/tmp/ks/**/*kernel_stats.csv',recursive=True)
tot=0
for r in list(csv.DictReader(open(f[0]))):
    calls=int(r['Calls']); 
    print("%-70s calls %6d avg us %9.1f total/step us %9.1f" % (r['Name'].replace('void amdspeech::','')[:70], calls, float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e3/10))
    tot+=float(r['TotalDurationNs'])/1e3/10
print("sum per step us", tot)
PY
tail -3 /tmp/ks.log
