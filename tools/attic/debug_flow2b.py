"""Dev tool (AMDSPEECH_DEVTRACE=3 build): check the two partial exchanges of lstm_bwd_flow2 against the GPU's OWN
gate gradients: dh_rec[t] = dG[t+1].W_hh^T (gathered P tiles) and dX_{l-1}[t] = dG_l[t].W_ih^T (gathered Q tiles)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
L, H, D, C, B, T, U = [int(v) for v in sys.argv[1].split(",")]
dbgbuf = torch.zeros(L * T * B * H, device="cuda")
os.environ["AMDSPEECH_TRACE_PTR"] = str(dbgbuf.data_ptr())
from rnn_speech_amd.engine import Engine
from rnn_speech_amd import lib as _l
eng = Engine(L, H, D, C, B, T, U, seed=7)
rng = np.random.RandomState(1)
x = rng.randn(T, B, D).astype(np.float32)
lengths = np.full(B, T, np.int32)
dense = np.zeros((B, U), np.int32); dense[:, :3] = rng.randint(1, C - 1, size=(B, 3)); dense[:, 3] = C - 1
eng.zero_grads()
eng.mini_batch(torch.as_tensor(x).cuda(), torch.as_tensor(lengths).cuda(), torch.as_tensor(dense).cuda())
torch.cuda.synchronize()
p = eng.to_numpy()
pad = lambda n: (n + 63) // 64 * 64
bp = (B + 15) // 16 * 16
tbh = T * B * H
sizes = [("wp", L * 2 * H * 4 * H), ("wq", L * 2 * H * 4 * H), ("z", (L + 1) * tbh), ("hs", L * (T + 1) * B * H), ("cs", L * (T + 1) * B * H),
         ("gates", L * tbh * 4), ("dg", L * tbh * 4), ("dztop", tbh), ("dz0", tbh), ("dc", L * 2 * B * H), ("xp0", T * bp * H),
         ("xp", L * 2 * bp * H), ("hp", L * 2 * bp * H), ("dgp", L * 2 * bp * 4 * H), ("sync", 64), ("xph", L * T * bp * H),
         ("hph", L * (T + 1) * bp * H), ("dxh", L * T * bp * H)]
off, offs = 0, {}
for name, n in sizes:
    offs[name] = off; off += pad(n)
assert eng._ws._offset(_l.WS_Z0) == offs["z"], (eng._ws._offset(_l.WS_Z0), offs["z"])
buf = eng._ws.buf
dg = buf[offs["dg"]:offs["dg"] + L * tbh * 4].view(L, T, B, 4 * H).cpu().numpy().astype(np.float64)
dxh = buf[offs["dxh"]:offs["dxh"] + L * T * bp * H].view(L, T, bp, H).cpu().numpy().astype(np.float64)
got = dbgbuf.view(L, T, B, H).cpu().numpy().astype(np.float64)
for l in range(L - 1, -1, -1):
    K = p["kernel_%d" % l].astype(np.float64)
    for t in range(T - 1, max(T - 5, -1), -1):
        msg = "layer %d t=%d:" % (l, t)
        if t + 1 < T:
            ref = dg[l, t + 1] @ K[H:, :].T
            err = np.abs(got[l, t] - ref); scale = np.abs(ref).max() + 1e-30
            per_cons = err.reshape(B, H // 16, 16).max(axis=(0, 2)) / scale
            per_row = err.max(axis=1) / scale
            msg += " P-sum max rel %.1e (bad consumers %s; bad rows %s)" % (err.max() / scale, ''.join('X' if v > 1e-3 else '.' for v in per_cons),
                                                                          ''.join('X' if v > 1e-3 else '.' for v in per_row))
        if l > 0:
            ref = dg[l, t] @ K[:H, :].T
            err = np.abs(dxh[l - 1, t, :B] - ref); scale = np.abs(ref).max() + 1e-30
            per_cons = err.reshape(B, H // 16, 16).max(axis=(0, 2)) / scale
            per_row = err.max(axis=1) / scale
            msg += " | Q-sum max rel %.1e (bad consumers %s; bad rows %s)" % (err.max() / scale, ''.join('X' if v > 1e-3 else '.' for v in per_cons),
                                                                            ''.join('X' if v > 1e-3 else '.' for v in per_row))
        print(msg)
