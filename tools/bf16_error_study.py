"""Round 4: what plain bf16 operands cost in accuracy over T ~ 1000 frames (precision = "bf16": one bf16 per value, one MFMA per
product, f32 accumulation / gates / state / master weights).  5x1024, D = 120, B = 64, T = 998, uni- and bidirectional, a live
pair of utterances against the float64 oracle -- beside bf16x3 and exact f32 on the same inputs.   python tools/bf16_error_study.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import model as om
from rnn_speech_amd.engine import Engine

L, H, D, C, B, T, U = 5, 1024, 120, 80, 64, 998, 161
if len(sys.argv) > 1 and sys.argv[1] == "cfg2":
    L, H, D, C, B, T, U = 3, 512, 40, 80, 32, 1001, 161


def rel(a, b):
    return float(np.abs(np.asarray(a, np.float64) - b).max() / (np.abs(b).max() + 1e-30))


for bidir in (False, True):
    ref = None
    for prec in ("f32", "bf16x3", "bf16"):
        eng = Engine(L, H, D, C, B, T, U, seed=99, precision=prec, bidirectional=bidir)
        rng = np.random.RandomState(12)
        p = eng.to_numpy()
        for k in p:
            if p[k].ndim == 1:
                p[k] = (rng.randn(*p[k].shape) * 0.1).astype(np.float32)
        eng.load_numpy(p)
        x = rng.randn(T, B, D).astype(np.float32)
        sel = [7, min(50, B - 1)]
        lengths = np.zeros(B, np.int32)
        lengths[sel[0]], lengths[sel[1]] = T - 85, T
        dense = np.zeros((B, U), np.int32)
        for b in range(B):
            n = rng.randint(80, 161)
            dense[b, :n - 1] = rng.randint(1, C - 1, size=n - 1)
            dense[b, n - 1] = C - 1
        if ref is None:
            p64 = {k: v.astype(np.float64) for k, v in p.items()}
            if bidir:
                logits_ref, cache = om.forward_bidirectional(p64, x[:, sel, :].astype(np.float64), lengths[sel], L)
            else:
                logits_ref, _, cache = om.forward(p64, x[:, sel, :].astype(np.float64), lengths[sel], L, keep_cache=True)
            loss_ref, dl = om.ctc_loss_and_grad(logits_ref, om.sparsify_labels(dense[sel], C), lengths[sel])
            g_ref = (om.backward_bidirectional if bidir else om.backward)(p64, cache, dl, lengths[sel], L)
            ref = (logits_ref, loss_ref, g_ref)
        logits_ref, loss_ref, g_ref = ref
        with eng.on_stream():
            eng.zero_grads()
            eng.mini_batch(torch.as_tensor(x).cuda(), torch.as_tensor(lengths).cuda(), torch.as_tensor(dense).cuda())
        torch.cuda.synchronize()
        eng.check()
        g = eng.to_numpy(eng.grads)
        gerr = {k: rel(g[k], g_ref[k]) for k in g_ref}
        worst = max(gerr, key=gerr.get)
        loss = eng.loss.cpu().numpy()[sel]
        ids_ok = om.greedy_decode(eng.logits.cpu().numpy()[:, sel, :].astype(np.float64), lengths[sel]) == om.greedy_decode(logits_ref, lengths[sel])
        print("%dx%d %s %-7s logits %.2e  loss rel %.2e  gradients: worst %.2e (%s), median %.2e  greedy ids identical: %s" % (
            L, H, "bidirectional " if bidir else "unidirectional", prec, rel(eng.logits.cpu().numpy()[:, sel, :], logits_ref),
            float(np.abs(loss - loss_ref).max() / np.abs(loss_ref).max()), gerr[worst], worst, float(np.median(list(gerr.values()))), ids_ok),
            flush=True)
        del eng
        torch.cuda.empty_cache()
