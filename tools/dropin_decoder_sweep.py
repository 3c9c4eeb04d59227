"""Dev tool: AcousticModel.run_train_step at configs[1] with the greedy GPU decoder and with the reference's beam decoder on host
threads (train_decoder_lag / decode threads per mini-batch): step time, duration of a decode job, CPU time the process burns per
step and how often the container's CPU quota throttled it.   python tools/dropin_decoder_sweep.py   (GPU)"""
import os, sys, time, resource
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from rnn_speech_amd import acoustic_model as am
stats = []
orig = am._AsyncBeamDecoder._decode
def timed_decode(self, *a):
    t0 = time.perf_counter(); r = orig(self, *a); stats.append(time.perf_counter() - t0); return r
am._AsyncBeamDecoder._decode = timed_decode
def cpu():
    r = resource.getrusage(resource.RUSAGE_SELF); return r.ru_utime + r.ru_stime
def throttled():
    try:
        return {l.split()[0]: int(l.split()[1]) for l in open("/sys/fs/cgroup/cpu.stat")}
    except OSError:
        return {}
def run(name, decoder):
    del stats[:]
    c0, th0, t0 = cpu(), throttled(), time.perf_counter()
    r = bench.dropin_run_train_step(40, decoder)
    c1, th1, t1 = cpu(), throttled(), time.perf_counter()
    job = " %.1f ms per decode job;" % (np.mean(stats[5:]) * 1e3) if stats else ""
    print("%-24s %.2f ms per step;%s %.1f host cores busy in the timed steps; throttled in %d of %d periods of the whole run"
          % (name, r["ms_per_step"], job, r["host_cores_busy"],
             th1.get("nr_throttled", 0) - th0.get("nr_throttled", 0), th1.get("nr_periods", 0) - th0.get("nr_periods", 0)), flush=True)
print("cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "?")
run("greedy", "greedy")
for lag, th in ((1, 16), (1, 12), (2, 16)):
    os.environ["AMDSPEECH_TRAIN_DECODER_LAG"] = str(lag); os.environ["AMDSPEECH_TRAIN_DECODER_THREADS"] = str(th)
    run("beam, lag %d, %2d threads" % (lag, th), "beam")
