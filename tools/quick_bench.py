"""Ad-hoc phase timing of one training mini-batch at a given config (dev tool)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rnn_speech_amd.engine import Engine
from rnn_speech_amd import ops

_nums = [v for v in sys.argv[1:] if not v.startswith('--')]
L, H, D, C, B, T, U = [int(v) for v in (_nums[:7] if len(_nums) >= 7 else (3, 512, 40, 80, 32, 1001, 161))]
eng = Engine(L, H, D, C, B, T, U, precision='bf16x3' if '--bf16x3' in sys.argv else 'f32')
rng = np.random.RandomState(0)
x = torch.as_tensor(rng.randn(T, B, D).astype(np.float32)).cuda()
lengths = torch.full((B,), T, dtype=torch.int32).cuda()
dense = np.zeros((B, U), np.int32)
for b in range(B):
    n = rng.randint(80, 161) if U > 160 else max(1, U // 2)
    dense[b, :n - 1] = rng.randint(1, C - 1, size=n - 1); dense[b, n - 1] = C - 1
dlab = torch.as_tensor(dense).cuda()

def timed(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.time() - t0) / n * 1e3

keep = (0.8, 0.5) if "--dropout" in sys.argv else (1.0, 1.0)
if "--stream" in sys.argv:          # run on a non-default (non-blocking) torch stream
    torch.cuda.synchronize()
    _st = torch.cuda.Stream()
    torch.cuda.set_stream(_st)
print("fwd  ms", timed(lambda: eng.forward(x, lengths, keep[0], keep[1], 1)))
print("ctc  ms", timed(lambda: eng.ctc(dlab, lengths)))
print("bwd  ms", timed(lambda: eng.backward(x, lengths)))
print("adam ms", timed(lambda: eng.apply(3e-4, 1.0)))
def step():
    eng.zero_grads(); eng.mini_batch(x, lengths, dlab, keep[0], keep[1], 1); eng.apply(3e-4, 1.0)
ms = timed(step, 5)
print("step ms", ms, "frames/s", B * T / ms * 1e3)
print("loss", eng.loss[:4].cpu().numpy())
