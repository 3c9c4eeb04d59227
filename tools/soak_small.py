import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from rnn_speech_amd.engine import Engine
from rnn_speech_amd import lib as _lib
sys.path.insert(0, "/root/repo/tests")
from test_gpu_model import make_batch
bad = 0
t0 = time.time()
for (L, H, B, T) in [(3, 128, 20, 40), (3, 512, 32, 24), (2, 256, 16, 33), (3, 384, 20, 17)]:
    D, C, U = 20, 80, 8
    for rep in range(6):
        eng = Engine(L, H, D, C, B, T, U, seed=7 + rep)
        batches = [tuple(torch.as_tensor(a).cuda() for a in make_batch(T, B, D, C, U, seed=40 + k)) for k in range(3)]
        with eng.on_stream():
            for i in range(60):
                eng.zero_grads()
                eng.mini_batch(*batches[i % 3], 0.8, 0.5, seed=i + 1)
                torch.cuda.synchronize()
                try:
                    eng.check()
                except _lib.DataflowTimeout as e:
                    bad += 1
                    print("TIMEOUT", (L, H, B, T), rep, i, str(e)[-200:][:120], flush=True)
        del eng
print("done bad=%d in %.1fs" % (bad, time.time() - t0))
