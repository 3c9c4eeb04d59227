"""Dev: per-tensor gradient error of the cfg2 full-size live-pair step against the float64 oracle, dropout off / on, f32 /
bf16x3 -- to tell f32 round-off (all tensors drift together, grows with depth) from a wrong mask element."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from oracle import model as om
from rnn_speech_amd.engine import Engine
from test_gpu_dropout_oracle import engine_masks, rel_err

L, H, D, C, B, T, U = 3, 512, 40, 80, 32, 1001, 161
rng = np.random.RandomState(3)
eng = Engine(L, H, D, C, B, T, U, seed=1234)
p = eng.to_numpy()
for k in p:
    if p[k].ndim == 1:
        p[k] = (rng.randn(*p[k].shape) * 0.1).astype(np.float32)
eng.load_numpy(p)
x = rng.randn(T, B, D).astype(np.float32)
sel = [3, 21]
live = np.zeros(B, np.int32); live[3], live[21] = T, 733
dense = np.zeros((B, U), np.int32)
for b in range(B):
    n = rng.randint(80, 161)
    dense[b, :n - 1] = rng.randint(1, C - 1, size=n - 1); dense[b, n - 1] = C - 1
dx, dlen, dlab = torch.as_tensor(x).cuda(), torch.as_tensor(live).cuda(), torch.as_tensor(dense).cuda()
p64 = {k: v.astype(np.float64) for k, v in p.items()}
for keep in ((1.0, 1.0), (0.8, 0.5), (0.8, 0.5), (0.8, 0.5)):
    seed = rng.randint(1 << 30)
    with eng.on_stream():
        eng.zero_grads()
        eng.mini_batch(dx, dlen, dlab, keep[0], keep[1], seed=seed)
    torch.cuda.synchronize(); eng.check()
    ins, outs = engine_masks(eng._ws, L)
    ins = [m[:, sel, :] for m in ins]; outs = [m[:, sel, :] for m in outs]
    lg, _, cache = om.forward(p64, x[:, sel, :].astype(np.float64), live[sel], L, keep_cache=True, in_masks=ins, out_masks=outs)
    loss_ref, dl = om.ctc_loss_and_grad(lg, om.sparsify_labels(dense[sel], C), live[sel])
    g_ref = om.backward(p64, cache, dl, live[sel], L, in_masks=ins, out_masks=outs)
    # f32 oracle of the same thing: how large is f32 round-off in this arithmetic on a CPU?
    p32 = {k: v.astype(np.float32) for k, v in p.items()}
    lg32, _, c32 = om.forward(p32, x[:, sel, :], live[sel], L, keep_cache=True, in_masks=[m.astype(np.float32) for m in ins],
                              out_masks=[m.astype(np.float32) for m in outs])
    _, dl32 = om.ctc_loss_and_grad(lg32, om.sparsify_labels(dense[sel], C), live[sel])
    g32 = om.backward(p32, c32, dl32, live[sel], L, in_masks=[m.astype(np.float32) for m in ins], out_masks=[m.astype(np.float32) for m in outs])
    g = eng.to_numpy(eng.grads)
    print("keep", keep, "logits", "%.2e" % rel_err(eng.logits.cpu().numpy()[:, sel], lg), "cpu-f32 %.2e" % rel_err(lg32, lg),
          "dlogits %.2e" % rel_err(eng.dlogits.cpu().numpy()[:, sel], dl), "cpu-f32 %.2e" % rel_err(dl32, dl))
    for k in g_ref:
        print("   %-10s gpu %.2e   cpu-f32 %.2e" % (k, rel_err(g[k], g_ref[k]), rel_err(g32[k], g_ref[k])))
