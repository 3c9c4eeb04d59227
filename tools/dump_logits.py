"""Dev tool: the logits the asynchronous training-time decoder sees in bench.py's drop-in run (a random-init 3x512 model on synthetic
speech, a few optimiser steps in) -> gpurun_out/logits_dropin.npy, for profiling csrc/beam.cpp on the host."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from rnn_speech_amd import acoustic_model as am
keep = {}
orig = am._AsyncBeamDecoder._decode
def grab(self, buf, done, Tr, lens, *a):
    r = orig(self, buf, done, Tr, lens, *a)
    keep["logits"] = buf[:Tr].numpy().copy(); keep["lens"] = np.asarray(lens).copy()
    return r
am._AsyncBeamDecoder._decode = grab
bench.dropin_run_train_step(8, "beam")
os.makedirs("gpurun_out", exist_ok=True)
np.save("gpurun_out/logits_dropin.npy", keep["logits"]); np.save("gpurun_out/lens_dropin.npy", keep["lens"])
lg = keep["logits"]; print(lg.shape, "logit std over labels (mean over frames): %.4f" % lg.std(axis=2).mean(), "std over everything %.4f" % lg.std())
