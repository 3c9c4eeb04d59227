"""Dev tool (build with AMDSPEECH_DEVTRACE=2): one step (t = 500) of every workgroup of the forward group
(layer 0, batch tile 0): when each passed its stamps, relative to the earliest."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
trace = torch.zeros(2 * 8 * 2 * 8, dtype=torch.int64, device="cuda")
os.environ["AMDSPEECH_TRACE_PTR"] = str(trace.data_ptr())
from rnn_speech_amd.engine import Engine
L, H, D, C, B, T, U = 3, 512, 40, 80, 32, 1001, 161
eng = Engine(L, H, D, C, B, T, U)
x = torch.randn(T, B, D, device="cuda"); lengths = torch.full((B,), T, dtype=torch.int32, device="cuda")
rng = np.random.RandomState(0)
dense = np.zeros((B, U), np.int32)
for b in range(B):
    n = rng.randint(80, 161); dense[b, :n - 1] = rng.randint(1, C - 1, size=n - 1); dense[b, n - 1] = C - 1
dlab = torch.as_tensor(dense).cuda()
for _ in range(3):
    eng.zero_grads(); eng.mini_batch(x, lengths, dlab)
torch.cuda.synchronize()
tr = trace.cpu().numpy().reshape(32, 8).astype(np.float64) / 100.0
t0 = tr[:, 0].min()
print(" ub |  loop top   h there  MFMAs out  partials   h sync   store out |  barrier B   loads out   (us)")
for ub in range(32):
    r = tr[ub] - t0
    print("%3d | %8.2f %9.2f %9.2f %9.2f %9.2f %9.2f  | %8.2f %9.2f" % (ub, r[0], r[1], r[2], r[3], r[4], r[7], r[5], r[6]))
