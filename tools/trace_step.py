"""Dev tool: per-wave s_memtime stamps of one forward diagonal of the launch-per-diagonal kernel
(AMDSPEECH_FLOW=0 AMDSPEECH_TRACE_PTR; build with AMDSPEECH_DEVTRACE=1).  tools/trace_flow.py traces the dataflow kernels."""
import os
os.environ.setdefault("AMDSPEECH_FLOW", "0")
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
trace = torch.zeros(4096 * 16 * 16, dtype=torch.int64, device="cuda")
os.environ["AMDSPEECH_TRACE_PTR"] = str(trace.data_ptr())
from rnn_speech_amd.engine import Engine
L, H, D, C, B, T, U = 3, 512, 40, 80, 32, 1001, 161
eng = Engine(L, H, D, C, B, T, U)
x = torch.randn(T, B, D, device="cuda"); lengths = torch.full((B,), T, dtype=torch.int32, device="cuda")
for _ in range(3):
    eng.forward(x, lengths)
torch.cuda.synchronize()
uw = int(os.environ.get("AMDSPEECH_UW", "8")); nw = int(os.environ.get("AMDSPEECH_FWD_NW", "8"))
nwg = (H // uw) * L
full = trace.cpu().numpy().reshape(-1, 16)[: nwg * nw]
tr = full[:, :8]
if int(os.environ.get('AMDSPEECH_DBG', '0')) & 16:
    arr = (full[:, 8:16] - full[:, 0:1]).astype(np.float64)
    print('arrival of K-block u after wave start (cycles): median per u', np.median(arr, axis=0).astype(int), ' max per u', arr.max(axis=0).astype(int))
    print('  min per u', arr.min(axis=0).astype(int), ' p10', np.percentile(arr, 10, axis=0).astype(int))
    print('  first WG, per wave block0/block7:', arr[:nw, 0].astype(int), arr[:nw, 7].astype(int))
    st0 = (full[:, 0] - full[:, 0].min()).astype(np.float64)
    print('  wave start spread (s_memtime, per-XCD clocks differ): first WG', st0[:nw].astype(int))
d = tr[:, :4].astype(np.float64)
ld = d[:, 1] - d[:, 0]; bar = d[:, 2] - d[:, 1]
ok = tr[:, 3] != 0
ep = (d[:, 3] - d[:, 2])[ok]
pr = lambda n, v: print("%-22s min %8.0f  med %8.0f  max %8.0f cycles" % (n, v.min(), np.median(v), v.max()))
pr("loads+mfma", ld); pr("barrier wait", bar); pr("epilogue (waves 0-3)", ep); pr("whole wave", (d[:, 3] - d[:, 0])[ok])
wc0 = tr[:, 7].astype(np.float64); wc1 = tr[ok, 6].astype(np.float64)
print("wall clock (100 MHz): first start -> last start %.2f us, first start -> last end %.2f us" % (
    (wc0.max() - wc0.min()) / 100.0, (wc1.max() - wc0.min()) / 100.0))
wave_us = (tr[ok, 6] - tr[ok, 7]).astype(np.float64) / 100.0
print("per-wave wall time: med %.2f us max %.2f us -> s_memtime ticks per us ~ %.0f" % (np.median(wave_us), wave_us.max(), np.median((d[:, 3] - d[:, 0])[ok]) / np.median(wave_us)))
