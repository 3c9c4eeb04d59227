"""Dev tool: wall time of AcousticModel.evaluate_full (per-file front end, forward, beam search width 100, WER / CER) on 64 synthetic
10 s utterances at the cfg2 model size."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from models.AcousticModel import AcousticModel, Session
from models.SpeechRecognizer import SpeechRecognizer
cm = SpeechRecognizer().get_char_map()
T, U, B = 1001, 161, 32
rng = np.random.RandomState(0)
words = ["hello", "there", "general", "speech", "recognition", "works", "on", "the", "new", "chip"]
items = [[((0.1 * rng.randn(160000)).astype(np.float32), 16000), " ".join(rng.choice(words, size=18)), None] for _ in range(64)]
model = AcousticModel(3, 512, B, T, U, 40, False, len(cm))
model.create_forward_rnn()
sess = Session()
model.evaluate_full(sess, items[:32], T, "mfcc", cm, n_mfcc=40, sample_rate=16000)
pr = cProfile.Profile(); t0 = time.time(); pr.enable()
wer, cer = model.evaluate_full(sess, items, T, "mfcc", cm, n_mfcc=40, sample_rate=16000)
pr.disable()
print("evaluate_full over 64 utterances: %.2f s (WER %.1f CER %.1f)" % (time.time() - t0, wer, cer))
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
