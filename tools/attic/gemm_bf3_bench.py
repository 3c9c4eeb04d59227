"""Dev: the split-precision GEMM against the exact-f32 kernels at the H = 1024 path's shapes (warm clocks)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rnn_speech_amd import ops
TB, H = 63872, 1024
shapes = [("x.W    [TB,H]x[H,4H]", (TB, 4 * H, H, False, False)), ("dX     [TB,4H]x[4H,H]^T", (TB, H, 4 * H, False, True)),
          ("dK     [TB,H]^Tx[TB,4H]", (H, 4 * H, TB, True, False)), ("dK 512 [32032,512]^Tx[32032,2048]", (512, 2048, 32032, True, False))]
def run(fn, a, b, ta, tb, out, n=5):
    for _ in range(3): fn(a, b, trans_a=ta, trans_b=tb, out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn(a, b, trans_a=ta, trans_b=tb, out=out)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
warm = torch.randn(4096, 4096, device="cuda")
for _ in range(30): ops.gemm(warm, warm)
for name, (M, N, K, ta, tb) in shapes:
    a = torch.randn((K, M) if ta else (M, K), device="cuda"); b = torch.randn((N, K) if tb else (K, N), device="cuda")
    out = torch.empty(M, N, device="cuda")
    t32 = run(ops.gemm, a, b, ta, tb, out); ref = out.clone()
    t3 = run(ops.gemm_bf16x3, a, b, ta, tb, out)
    fl = 2.0 * M * N * K
    print("%-36s f32 %.3f ms (%.0f TF)   bf16x3 %.3f ms (%.0f TF-equivalent)   rel diff %.1e" % (
        name, t32 * 1e3, fl / t32 / 1e12, t3 * 1e3, fl / t3 / 1e12, float((out - ref).abs().max() / ref.abs().max())))
