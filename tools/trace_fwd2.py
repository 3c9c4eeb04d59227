"""Dev tool (build with AMDSPEECH_DEVTRACE=5): wall-clock stamps of lstm_fwd_flow2, layer 1, unit block 3, waves 0 (epilogue) and 5,
steps 500..507 -- per-phase microseconds."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
trace = torch.zeros(8 * 2 * 8, dtype=torch.int64, device="cuda")
os.environ["AMDSPEECH_TRACE_PTR"] = str(trace.data_ptr())
from rnn_speech_amd.engine import Engine
L, H, D, C, B, T, U = 3, 512, 40, 80, 32, 1001, 161
eng = Engine(L, H, D, C, B, T, U)
x = torch.randn(T, B, D, device="cuda"); lengths = torch.full((B,), T, dtype=torch.int32, device="cuda")
for _ in range(3):
    eng.forward(x, lengths, 0.8, 0.5, 1)
torch.cuda.synchronize()
tr = trace.cpu().numpy().reshape(8, 2, 8).astype(np.float64) / 100.0
names = ["settle h", "h MFMA+LDS", "B1", "epilogue|stores", "B2", "settle x", "x MFMA"]
for w, nm in ((0, "wave 0 (epilogue wave)"), (1, "wave 5 (store wave)")):
    print(nm)
    print("   t  " + "  ".join("%15s" % n for n in names) + " |  period")
    for i in range(8):
        r = tr[i, w]
        per = tr[i + 1, w, 0] - r[0] if i < 7 else float("nan")
        print("%4d  " % (500 + i) + "  ".join("%15.2f" % (r[k + 1] - r[k]) for k in range(7)) + " | %7.2f" % per)
