"""Dev tool: the same cfg2 mini-batch N times; the loss must repeat bit for bit (no atomics on its path), the gradients within the
f32 atomics' reordering noise.  A race in a hand-off (a stale tile, a fragment copied in flight) shows up as an outlier."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rnn_speech_amd.engine import Engine
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
L, H, D, C, B, T, U = 3, 512, 40, 80, 32, 1001, 161
if len(sys.argv) > 2 and sys.argv[2] == "big":      # the per-layer H = 1024 kernels (lstm_fwd_big / lstm_bwd_big), all four batch tiles
    L, H, D, C, B, T, U = 2, 1024, 120, 80, 64, 400, 60
prec = sys.argv[3] if len(sys.argv) > 3 else "f32"
eng = Engine(L, H, D, C, B, T, U, precision=prec)
rng = np.random.RandomState(0)
x = torch.as_tensor(rng.randn(T, B, D).astype(np.float32)).cuda()
lengths = torch.as_tensor(rng.randint(T * 6 // 10, T + 1, size=B).astype(np.int32)).cuda()
dense = np.zeros((B, U), np.int32)
for b in range(B):
    n = rng.randint(U // 2, U - 1); dense[b, :n - 1] = rng.randint(1, C - 1, size=n - 1); dense[b, n - 1] = C - 1
dlab = torch.as_tensor(dense).cuda()
torch.cuda.set_stream(eng.stream)
ref_loss = ref_g = None
worst = 0.0
for i in range(N):
    eng.zero_grads(); eng.mini_batch(x, lengths, dlab, 0.8, 0.5, 7, max_len=int(lengths.max()))
    torch.cuda.synchronize(); eng.check()
    if ref_loss is None:
        ref_loss, ref_g = eng.loss.clone(), eng.grads.clone()
        continue
    assert torch.equal(eng.loss, ref_loss), ("loss differs at iteration", i, float((eng.loss - ref_loss).abs().max()))
    d = float((eng.grads - ref_g).abs().max() / ref_g.abs().max())
    worst = max(worst, d)
    assert d < 1e-5, ("gradient outlier at iteration", i, d)
print("%d iterations: loss bit-identical, gradients within %.1e of the first run" % (N, worst))
