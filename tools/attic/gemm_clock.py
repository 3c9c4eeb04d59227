import os, sys
sys.path.insert(0, "/root/repo")
import torch
from rnn_speech_amd import ops
def run(M,N,K,reps):
    a = torch.randn(K, M, device="cuda"); b = torch.randn(K, N, device="cuda"); out = torch.zeros(M, N, device="cuda")
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    evs[0].record()
    for i in range(reps):
        ops.gemm(a, b, trans_a=True, out=out, accumulate=True)
        evs[i + 1].record()
    torch.cuda.synchronize()
    ts = [evs[i].elapsed_time(evs[i + 1]) for i in range(reps)]
    print(M, N, K, " ".join("%.3f" % t for t in ts[:: max(1, reps // 16)]))
run(512, 2048, 23063, 200)
run(1024, 4096, 64064, 16)
run(512, 2048, 23063, 32)
