import os, sys, subprocess, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from rnn_speech_amd.engine import Engine
L, H, D, C, B, T, U = [int(v) for v in os.environ.get("CMP_CFG", "3,512,40,80,32,1001,161").split(",")]
rng = np.random.RandomState(0)
x = torch.as_tensor(rng.randn(T, B, D).astype(np.float32)).cuda()
lengths = torch.as_tensor(rng.randint(T // 2, T + 1, size=B).astype(np.int32)).cuda()
dense = np.zeros((B, U), np.int32)
for b in range(B):
    n = rng.randint(2, max(3, min(U, T // 3))); dense[b, :n - 1] = rng.randint(1, C - 1, size=n - 1); dense[b, n - 1] = C - 1
dlab = torch.as_tensor(dense).cuda()
eng = Engine(L, H, D, C, B, T, U, seed=1234)
eng.zero_grads(); eng.mini_batch(x, lengths, dlab); torch.cuda.synchronize(); eng.check()
g = eng.to_numpy(eng.grads)
np.savez(sys.argv[1], **g)
