"""Times the f32 GEMM entry (ops.gemm) at the hot path's big shapes.   python tools/gemm_f32_bench.py   (GPU)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rnn_speech_amd import ops

SHAPES = [("dZ_0 cfg2 (NT)", 32032, 512, 2048, False, True), ("dX cfg3 (NT)", 63872, 1024, 4096, False, True),
          ("x.W cfg3 (NN)", 63872, 4096, 1024, False, False), ("x.W cfg3 (NT)", 63872, 4096, 1024, False, True), ("x.W K=2048 (NN)", 63872, 4096, 2048, False, False),
          ("dK cfg2 (TN)", 512, 2048, 19860, True, False), ("dK cfg3 (TN)", 1024, 4096, 63872, True, False)]
for name, M, N, K, ta, tb in SHAPES:
    a = torch.randn((K, M) if ta else (M, K), device="cuda")
    b = torch.randn((N, K) if tb else (K, N), device="cuda")
    out = torch.empty(M, N, device="cuda")
    for _ in range(3):
        ops.gemm(a, b, trans_a=ta, trans_b=tb, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        ops.gemm(a, b, trans_a=ta, trans_b=tb, out=out)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    print("%-18s M=%d N=%d K=%d: %8.1f us  %6.1f TFLOP/s" % (name, M, N, K, us, 2.0 * M * N * K / us / 1e6), flush=True)
