"""Dev tool: TF/s of the f32 MFMA GEMM (csrc/gemm.hip) at the hot path's shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rnn_speech_amd import ops

def bench(name, M, N, K, ta, tb, reps=10, acc=False):
    a = torch.randn((K, M) if ta else (M, K), device="cuda")
    b = torch.randn((N, K) if tb else (K, N), device="cuda")
    out = torch.zeros(M, N, device="cuda")
    for _ in range(max(2, int(30e-3 / (2.0 * M * N * K / 100e12)))):      # ~30 ms: lets the clocks ramp up
        ops.gemm(a, b, trans_a=ta, trans_b=tb, out=out, accumulate=acc)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.gemm(a, b, trans_a=ta, trans_b=tb, out=out, accumulate=acc)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    if acc:
        out.zero_(); ops.gemm(a, b, trans_a=ta, trans_b=tb, out=out, accumulate=True)
    ref = (a.t() if ta else a).double() @ (b.t() if tb else b).double()
    err = float((out.double() - ref).abs().max() / ref.abs().max())
    print("%-34s M %5d N %5d K %6d  %8.3f ms  %6.1f TF/s   rel err %.1e" % (name, M, N, K, ms, 2.0 * M * N * K / ms / 1e9, err))

SHAPES = [
    ("cfg2 dK half, 72% of frames", 512, 2048, 23063, True, False),
    ("cfg2 dK half, all frames", 512, 2048, 32032, True, False),
    ("cfg2 dZ0 = dG0 Kx^T", 23063, 512, 2048, False, True),
    ("cfg3 x.W_ih hoist", 64064, 4096, 1024, False, False),
    ("cfg3 dK half", 1024, 4096, 64064, True, False),
    ("cfg3 dX = dG Kx^T", 64064, 1024, 4096, False, True),
    ("dense out fwd z3 W2", 32032, 80, 512, False, False),
    ("dense out bwd dlogits W2^T", 32032, 512, 80, False, True),
    ("dense in fwd x W1", 32032, 512, 40, False, False),
    ("square 4096", 4096, 4096, 4096, False, False),
]
only = [a for a in sys.argv[1:] if "," not in a]
for a in sys.argv[1:]:
    if "," in a:                     # custom shape: M,N,K,ta,tb
        f = [int(v) for v in a.split(",")]
        m, n, k, ta, tb = f[:5]
        bench("custom acc" if len(f) > 5 and f[5] else "custom", m, n, k, bool(ta), bool(tb), reps=10, acc=len(f) > 5 and bool(f[5]))
if any("," in a for a in sys.argv[1:]) and not only:
    sys.exit(0)
for sh in SHAPES:
    if not only or any(o in sh[0] for o in only):
        bench(*sh, reps=5 if sh[1] * sh[2] * sh[3] > 1e11 else 10)
