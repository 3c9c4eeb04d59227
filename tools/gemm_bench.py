"""Dev tool: TF/s of the f32 MFMA GEMM (csrc/gemm.hip) at the hot path's shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rnn_speech_amd import ops

def bench(name, M, N, K, ta, tb, reps=10):
    a = torch.randn((K, M) if ta else (M, K), device="cuda")
    b = torch.randn((N, K) if tb else (K, N), device="cuda")
    out = torch.zeros(M, N, device="cuda")
    for _ in range(2):
        ops.gemm(a, b, trans_a=ta, trans_b=tb, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.gemm(a, b, trans_a=ta, trans_b=tb, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    ref = (a.t() if ta else a).double() @ (b.t() if tb else b).double()
    err = float((out.double() - ref).abs().max() / ref.abs().max())
    print("%-34s M %5d N %5d K %6d  %8.3f ms  %6.1f TF/s   rel err %.1e" % (name, M, N, K, ms, 2.0 * M * N * K / ms / 1e9, err))

bench("weight grad dK = [z;h]^T dG", 1024, 2048, 32032, True, False)
bench("weight grad, 1/4 of the frames", 1024, 2048, 8008, True, False)
bench("dense out fwd z3 W2", 32032, 80, 512, False, False)
bench("dense out bwd dlogits W2^T", 32032, 512, 80, False, True)
bench("dense in fwd x W1", 32032, 512, 40, False, False)
bench("square 4096", 4096, 4096, 4096, False, False)
