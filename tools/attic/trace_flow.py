"""Dev tool (build with AMDSPEECH_DEVTRACE=1): wall-clock stamps of the dataflow forward kernel, layer 1,
unit block 3, waves 0 (epilogue + K slice) and 5 (K slice only), steps 500..507."""
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
both = trace.cpu().numpy().reshape(2, 8, 2, 8).astype(np.float64) / 100.0
tr = both[0]
names = ["settle", "MFMAs", "write partials", "barrier A", "epilogue + barrier B", "-"]
for w, nm in ((0, "wave 0 (h wave: h half of step t + epilogue)"), (1, "wave 5 (x wave: x half of step t+1)")):
    print(nm)
    print("   t  " + "  ".join("%15s" % n for n in names[:5]) + " |  period")
    for i in range(8):
        r = tr[i, w]
        per = tr[i + 1, w, 0] - r[0] if i < 7 else float("nan")
        print("%4d  " % (500 + i) + "  ".join("%15.2f" % (r[k + 1] - r[k]) for k in range(5)) + " | %7.2f" % per
              + ("   issue %.2f, to next top %.2f" % (r[6] - r[5], tr[i + 1, w, 0] - r[6]) if w == 0 and i < 7 else ""))

tb = both[1]
names = ["settle dG[t+1]", "rec MFMAs", "red+barrier", "epilogue+stores", "down MFMAs", "issue+barrier+dX"]
print("BACKWARD (steps listed in execution order: t = 507 .. 500)")
for w, nm in ((0, "wave 0 (epilogue)"), (1, "wave 5")):
    print(nm)
    print("   t  " + "  ".join("%15s" % n for n in names) + " |  period")
    for i in range(7, -1, -1):
        r = tb[i, w]
        per = tb[i - 1, w, 0] - r[0] if i > 0 else float("nan")
        print("%4d  " % (500 + i) + "  ".join("%15.2f" % (r[k + 1] - r[k]) for k in range(6)) + " | %7.2f" % per)
