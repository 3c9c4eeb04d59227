import os, sys, time, cProfile, pstats
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from models.AcousticModel import AcousticModel, Session
from models.SpeechRecognizer import SpeechRecognizer
cm = SpeechRecognizer().get_char_map()
T, U, B = 1001, 161, 32
rng = np.random.RandomState(0)
words = ["hello", "there", "general", "speech", "recognition", "works", "on", "the", "new", "chip"]
items = []
for i in range(B * 14):
    sig = (0.1 * rng.randn(160000)).astype(np.float32)
    items.append([(sig, 16000), " ".join(rng.choice(words, size=18)), None])
model = AcousticModel(3, 512, B, T, U, 40, False, len(cm))
sess = Session()
ds = model.build_dataset(items, B, T, U, "mfcc", cm, n_mfcc=40)
t_it, v_it = model.add_datasets_input(ds, model.build_dataset(items[:B], B, T, U, "mfcc", cm, n_mfcc=40))
sess.run(t_it.initializer); sess.run(v_it.initializer)
model.create_training_rnn(0.8, 0.5, 1, 3e-4, 0.33, use_iterator=True)
for _ in range(3): model.run_train_step(sess, 1, 1.0)
torch.cuda.synchronize()
pr = cProfile.Profile(); t0 = time.time(); pr.enable()
for _ in range(10): model.run_train_step(sess, 1, 1.0)
torch.cuda.synchronize(); pr.disable()
print("%.2f ms per run_train_step" % ((time.time() - t0) / 10 * 1e3))
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
