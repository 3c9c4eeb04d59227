"""Dev: exposed cost of the asynchronous training-time beam decoder in the drop-in step, per pipeline depth."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from rnn_speech_amd.engine import Engine
for lag in [int(v) for v in sys.argv[1:]] or [2, 4, 6]:
    os.environ["AMDSPEECH_TRAIN_DECODER_LAG"] = str(lag)
    r = bench.dropin_run_train_step(24, train_decoder="beam")
    print("lag %d: %.2f ms per step  err %.3f" % (lag, r["ms_per_step"], r["last_error_rate"]))
r = bench.dropin_run_train_step(24)
print("greedy: %.2f ms per step" % r["ms_per_step"])
