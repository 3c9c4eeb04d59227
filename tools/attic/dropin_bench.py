"""Dev tool: time the drop-in AcousticModel.run_train_step (dataset -> features -> step -> error rate)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from models.AcousticModel import AcousticModel, Session
from models.SpeechRecognizer import SpeechRecognizer
cm = SpeechRecognizer().get_char_map()
T, U, B = 1001, 161, 32
rng = np.random.RandomState(0)
words = ["hello", "there", "general", "speech", "recognition", "works", "on", "the", "new", "chip"]
items = []
for i in range(B * 6):
    sig = (0.1 * rng.randn(160000)).astype(np.float32)
    txt = " ".join(rng.choice(words, size=18))
    items.append([(sig, 16000), txt, None])
model = AcousticModel(3, 512, B, T, U, 40, False, len(cm))
sess = Session()
ds = model.build_dataset(items, B, T, U, "mfcc", cm, n_mfcc=40)
t_it, v_it = model.add_datasets_input(ds, model.build_dataset(items[:B], B, T, U, "mfcc", cm, n_mfcc=40))
sess.run(t_it.initializer); sess.run(v_it.initializer)
model.create_training_rnn(0.8, 0.5, 1, 3e-4, 0.33, use_iterator=True)
for flag in (True, False):
    model.compute_error_rate = flag
    sess.run(t_it.initializer)
    model.run_train_step(sess, 1, 1.0)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(4):
        loss, err, step, empty = model.run_train_step(sess, 1, 1.0)
    torch.cuda.synchronize()
    print("compute_error_rate=%s: %.1f ms per run_train_step (loss %.3f err %.3f)" % (flag, (time.time() - t0) / 4 * 1e3, loss, err))
