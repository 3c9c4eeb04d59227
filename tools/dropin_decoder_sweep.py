"""Dev tool: AcousticModel.run_train_step at configs[1] with the greedy GPU decoder and with the reference's beam decoder on host
threads at several pipeline depths (train_decoder_lag); where a decode job's time goes.   python tools/dropin_decoder_sweep.py   (GPU)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from rnn_speech_amd import acoustic_model as am, ops
stats = []
orig = am._AsyncBeamDecoder._decode
def timed_decode(self, *a):
    t0 = time.perf_counter(); r = orig(self, *a); stats.append(time.perf_counter() - t0); return r
am._AsyncBeamDecoder._decode = timed_decode
r = bench.dropin_run_train_step(20, "greedy")
print("greedy           %.2f ms per step" % r["ms_per_step"], flush=True)
for lag, th in ((2, 32), (2, 12), (1, 32), (1, 16), (0, 0)):
    os.environ["AMDSPEECH_TRAIN_DECODER_LAG"] = str(lag); os.environ["AMDSPEECH_TRAIN_DECODER_THREADS"] = str(th)
    del stats[:]
    r = bench.dropin_run_train_step(20, "beam")
    print("beam, lag %d, %2d threads   %.2f ms per step   (last error rate %.3f); %.1f ms per decode job"
          % (lag, th, r["ms_per_step"], r["last_error_rate"], np.mean(stats[5:]) * 1e3), flush=True)
