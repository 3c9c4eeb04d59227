"""Dev tool (build with AMDSPEECH_DEVTRACE=5): wall-clock stamps of lstm_fwd_flow2, layer 1, unit block 3, waves 0 (epilogue) and 5,
steps 500..507 -- per-phase microseconds."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
trace = torch.zeros(8 * 2 * 8 + 64, dtype=torch.int64, device="cuda")
os.environ["AMDSPEECH_TRACE_PTR"] = str(trace.data_ptr())
from rnn_speech_amd.engine import Engine
L, H, D, C, B, T, U = 3, 512, 40, 80, 32, 1001, 161
eng = Engine(L, H, D, C, B, T, U)
x = torch.randn(T, B, D, device="cuda"); lengths = torch.full((B,), T, dtype=torch.int32, device="cuda")
for _ in range(3):
    eng.forward(x, lengths, 0.8, 0.5, 1)
torch.cuda.synchronize()
raw = trace.cpu().numpy().astype(np.float64) / 100.0
tr = raw[:128].reshape(8, 2, 8)
wk = raw[128:].reshape(8, 8)
names = ["settle h", "h MFMA+LDS", "B1", "epilogue|stores", "B2", "settle x", "x MFMA"]
for w, nm in ((0, "wave 0 (epilogue wave)"), (1, "wave 5 (store wave)")):
    print(nm)
    print("   t  " + "  ".join("%15s" % n for n in names) + " |  period")
    for i in range(8):
        r = tr[i, w]
        per = tr[i + 1, w, 0] - r[0] if i < 7 else float("nan")
        print("%4d  " % (500 + i) + "  ".join("%15.2f" % (r[k + 1] - r[k]) for k in range(7)) + " | %7.2f" % per)

if wk.any():
    print("x-product worker (layer 1, unit block 3, part 0): microseconds")
    print("   t      wait(12)   checks    MFMAs+stores   loads |  period   (start - consumer's step start of the same t)")
    for i in range(8):
        r = wk[i]
        per = wk[i + 1, 0] - r[0] if i < 7 else float("nan")
        print("%4d  %9.2f %9.2f %12.2f %9.2f | %7.2f   %9.2f" % (500 + i, r[1] - r[0], r[2] - r[1], r[3] - r[2], r[4] - r[3], per, r[0] - tr[i, 0, 0]))
