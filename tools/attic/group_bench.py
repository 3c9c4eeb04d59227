"""Dev tool: the grouped weight-gradient GEMM launch at cfg2 (6 problems 512 x 2048 x K) through Engine-free calls."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rnn_speech_amd import ops
K = int(sys.argv[1]) if len(sys.argv) > 1 else 23063
n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
a = [torch.randn(K, 512, device="cuda") for _ in range(n)]
b = [torch.randn(K, 2048, device="cuda") for _ in range(n)]
c = [torch.zeros(512, 2048, device="cuda") for _ in range(n)]
def run():
    for i in range(n):
        ops.gemm(a[i], b[i], trans_a=True, out=c[i], accumulate=True)
for _ in range(20): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print("separate launches: %d GEMMs %.3f ms  %.1f TF/s" % (n, ms, n * 2.0 * 512 * 2048 * K / ms / 1e9))
