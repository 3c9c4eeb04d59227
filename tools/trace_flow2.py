"""Dev tool (build with AMDSPEECH_DEVTRACE=1): wall-clock stamps of lstm_bwd_flow2, layer 1, unit block 3, waves 0 and 5,
steps 507..500 (execution order).  Stamps cost ~5 %."""
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
torch.cuda.set_stream(eng.stream)
import ctypes
from rnn_speech_amd import lib as _lib
lib = _lib.load(); lib.amdspeech_profile_enable(1)
keep = (0.8, 0.5) if os.environ.get("TRACE_DROPOUT") == "1" else (1.0, 1.0)
for i in range(3):
    eng.zero_grads(); eng.mini_batch(x, lengths, dlab, keep[0], keep[1], i + 1)
torch.cuda.synchronize()
ms, nl = ctypes.c_float(), ctypes.c_int()
lib.amdspeech_profile_get(1, ctypes.byref(ms), ctypes.byref(nl))
print("layer %s  keep %s  bwd kernel %.3f ms = %.2f us per time step" % (os.environ.get("AMDSPEECH_TRACE_LAYER", "top"), keep, ms.value, ms.value * 1e3 / nl.value))
tb = trace.cpu().numpy().reshape(2, 8, 2, 8).astype(np.float64)[1] / 100.0
names = ["settle P+red_r", "barrier B1", "epilogue|dX", "barrier B2", "Qissue+rec MFMA", "P store", "down MFMA+Q st"]
if os.environ.get("TRACE_SET") == "4":
    names = ["top..B2", "Qiss+rec issue", "P store", "prefetch|rowmaj", "down 1st part", "gather issue", "down 2nd part"]
names = ["settle P+red_r", "barrier B1", "epilogue|dX,rowmaj", "barrier B2", "Qissue+rec MFMA", "P store", "down MFMA+Q st"]
for w, nm in ((0, "wave 0 (epilogue wave)"), (1, "wave 5 (dX / row-major wave)")):
    print(nm)
    print("   t  " + "  ".join("%16s" % n for n in names) + " |  period")
    for i in range(7, -1, -1):
        r = tb[i, w]
        per = tb[i - 1, w, 0] - r[0] if i > 0 else float("nan")
        print("%4d  " % (500 + i) + "  ".join("%16.2f" % (r[k + 1] - r[k]) for k in range(7)) + " | %7.2f" % per)
