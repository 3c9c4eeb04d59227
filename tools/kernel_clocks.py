"""Dev tool: effective shader clock of the two dataflow recurrence kernels INSIDE the training step vs run alone
   (s_memtime / s_memrealtime deltas written by the kernels when AMDSPEECH_TRACE_PTR points at a device buffer)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
buf = torch.zeros(64, dtype=torch.int64, device="cuda")
os.environ["AMDSPEECH_TRACE_PTR"] = str(buf.data_ptr())
from rnn_speech_amd.engine import Engine
L, H, D, C, B, T, U = 3, 512, 40, 80, 32, 1001, 161
eng = Engine(L, H, D, C, B, T, U)
rng = np.random.RandomState(0)
x = torch.as_tensor(rng.randn(T, B, D).astype(np.float32)).cuda()
lengths = torch.full((B,), T, dtype=torch.int32).cuda()
dense = np.zeros((B, U), np.int32)
for b in range(B):
    n = 120; dense[b, :n - 1] = rng.randint(1, C - 1, size=n - 1); dense[b, n - 1] = C - 1
dlab = torch.as_tensor(dense).cuda()
torch.cuda.set_stream(eng.stream)
def report(tag):
    v = buf.cpu().numpy()
    f = 100.0 * v[0] / max(1, v[1]); b = 100.0 * v[2] / max(1, v[3])
    print("%-28s fwd %.3f ms @ %4.0f MHz   bwd %.3f ms @ %4.0f MHz" % (tag, v[1] / 1e5, f, v[3] / 1e5, b))
for rep in range(2):
    for _ in range(60): eng.forward(x, lengths, 0.8, 0.5, 1)
    torch.cuda.synchronize(); report("forward only, 60 in a row")
    for _ in range(60):
        eng.zero_grads(); eng.mini_batch(x, lengths, dlab, 0.8, 0.5, 1); eng.apply(1e-4, 1.0)
    torch.cuda.synchronize(); report("whole steps, 60 in a row")
