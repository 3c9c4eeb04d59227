# gpurun -- bash tools/bench_fwd.sh: the bench step's recurrence times with the front end beside the forward kernel / beside the CTC stage / excluded
one() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step %.3f fwd %.3f bwd %.3f' % (d['ms_per_step'], d['config']['fwd_recurrence_ms'], d['config']['bwd_recurrence_ms']))"; }
echo "default:            $(one)"
echo "beside CTC:         $(AMDSPEECH_BESIDE_FORWARD=0 one)"
echo "no front end:       $(one --no-frontend)"
echo "workers off:        $(AMDSPEECH_FLOW_FWD_WORKERS=0 one)"
echo "default again:      $(one)"
