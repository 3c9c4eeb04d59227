"""Times the dense layers' products (output Linear, its input gradient, input Linear) on the short-axis kernels of
csrc/gemm_skinny.hip and, with AMDSPEECH_GEMM_SKINNY=0 in a second process, on the general LDS kernel.
    python tools/gemm_skinny_bench.py            (GPU)"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    import torch
    from rnn_speech_amd import ops
    shapes = [("output Linear cfg2", 32032, 80, 512, False), ("dztop cfg2", 32032, 512, 80, True),
              ("input Linear cfg2", 32032, 512, 40, False), ("output Linear cfg3", 63872, 80, 1024, False),
              ("dztop cfg3", 63872, 1024, 80, True), ("input Linear cfg3", 63872, 1024, 120, False)]
    for name, M, N, K, tb in shapes:
        a = torch.randn(M, K, device="cuda")
        b = torch.randn(N, K, device="cuda") if tb else torch.randn(K, N, device="cuda")
        bias = torch.randn(N, device="cuda")
        out = torch.empty(M, N, device="cuda")
        for _ in range(3):
            ops.gemm(a, b, trans_b=tb, bias=bias, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 50
        e0.record()
        for _ in range(n):
            ops.gemm(a, b, trans_b=tb, bias=bias, out=out)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        print("%-22s M=%d N=%d K=%d: %7.1f us  %6.1f TFLOP/s  %5.2f TB/s" %
              (name, M, N, K, us, 2.0 * M * N * K / us / 1e6, 4.0 * (M * K + M * N + N * K) / us / 1e6), flush=True)
    # weight + bias gradients (linear_bwd without dx): rows x K_in x N_out
    for name, M, K, N in [("dW_in cfg2", 32032, 40, 512), ("dW_out cfg2", 32032, 512, 80), ("dW_in cfg3", 63872, 120, 1024),
                          ("dW_out cfg3", 63872, 1024, 80)]:
        x, dy, w = torch.randn(M, K, device="cuda"), torch.randn(M, N, device="cuda"), torch.randn(K, N, device="cuda")
        dw, db = torch.zeros(K, N, device="cuda"), torch.zeros(N, device="cuda")
        for _ in range(3):
            ops.linear_bwd(x, w, dy, dw, db, need_dx=False)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 50
        e0.record()
        for _ in range(n):
            ops.linear_bwd(x, w, dy, dw, db, need_dx=False)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        print("%-22s rows=%d %dx%d: %7.1f us  %6.1f TFLOP/s  %5.2f TB/s" %
              (name, M, K, N, us, 2.0 * M * N * K / us / 1e6, 4.0 * (M * K + M * N) / us / 1e6), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run()
    else:
        for v in ("1", "0"):
            print("AMDSPEECH_GEMM_SKINNY=" + v, flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, AMDSPEECH_GEMM_SKINNY=v))
