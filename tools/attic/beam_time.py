import sys, time; sys.path.insert(0, "/root/repo")
import numpy as np, torch
from rnn_speech_amd import ops
T, B, C = 1001, 32, 80
rng = np.random.RandomState(0)
logits = torch.as_tensor((rng.randn(T, B, C) * 3).astype(np.float32)).cuda()
lengths = torch.full((B,), T, dtype=torch.int32).cuda()
for w in (1, 10, 100):
    ops.ctc_beam_search(logits, lengths, w, True)
    t0 = time.time(); ops.ctc_beam_search(logits, lengths, w, True); print("beam width %3d: %.1f ms per batch of 32 x 1001 frames" % (w, (time.time() - t0) * 1e3))
# peaky (trained-like) distribution
peaky = logits.clone(); peaky[:, :, C - 1] += 8.0
t0 = time.time(); ops.ctc_beam_search(peaky, lengths, 100, True); print("peaky, width 100: %.1f ms" % ((time.time() - t0) * 1e3))
