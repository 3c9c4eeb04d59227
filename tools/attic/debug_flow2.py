"""Dev tool: per-(layer, frame) error of the backward recurrence's gate gradients against the float64 oracle."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import model as om
from rnn_speech_amd.engine import Engine
from rnn_speech_amd.lib import AmdSpeechError
for cfg in sys.argv[1:]:
    L, H, D, C, B, T, U = [int(v) for v in cfg.split(",")]
    eng = Engine(L, H, D, C, B, T, U, seed=7)
    rng = np.random.RandomState(1)
    x = rng.randn(T, B, D).astype(np.float32)
    lengths = np.full(B, T, np.int32)
    dense = np.zeros((B, U), np.int32); dense[:, :3] = rng.randint(1, C - 1, size=(B, 3)); dense[:, 3] = C - 1
    p64 = {k: v.astype(np.float64) for k, v in eng.to_numpy().items()}
    logits, _, cache = om.forward(p64, x.astype(np.float64), lengths, L, keep_cache=True)
    _, dl = om.ctc_loss_and_grad(logits, om.sparsify_labels(dense, C), lengths)
    dbg = {}
    om.backward(p64, cache, dl, lengths, L, debug=dbg)
    eng.zero_grads()
    eng.mini_batch(torch.as_tensor(x).cuda(), torch.as_tensor(lengths).cuda(), torch.as_tensor(dense).cuda())
    torch.cuda.synchronize()
    try:
        eng.check(); status = "ok"
    except AmdSpeechError as exc:
        status = "TIMEOUT"
    ws = eng._ws
    # dg region: right after gates in the layout; reach it through the descriptor's own pointer arithmetic
    import ctypes as Cc
    from rnn_speech_amd import lib as _l
    z0 = ws._offset(_l.WS_Z0)
    tbh = T * B * H
    off_dg = z0 + (L + 1) * tbh + 2 * L * (T + 1) * B * H + L * tbh * 4
    def pad(n): return (n + 63) // 64 * 64
    # recompute with the 64-float padding of every region
    off = z0
    for n in ((L + 1) * tbh, L * (T + 1) * B * H, L * (T + 1) * B * H, L * tbh * 4):
        off += pad(n)
    dg = ws.buf[off:off + L * tbh * 4].view(L, T, B, 4 * H).cpu().numpy()
    print(cfg, status)
    for l in range(L - 1, -1, -1):
        ref = dbg["dg_%d" % l]
        err = np.abs(dg[l] - ref).reshape(T, -1).max(axis=1) / (np.abs(ref).max() + 1e-30)
        bad = np.where(err > 1e-3)[0]
        print("  layer %d: max rel err %.2e; frames over 1e-3: %s" % (l, err.max(), (list(bad[-6:][::-1]) if len(bad) else "none")),
              " err by frame (last 8):", " ".join("%.1e" % v for v in err[::-1][:8]))
