"""Dev check: one rank, RCCL initialised, an all-reduce of the flat gradient buffer every step on the engine's
stream -- does the RCCL stream disturb the CU-partitioned backward overlap?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1)
from rnn_speech_amd.engine import Engine
L, H, D, C, B, T, U = 3, 512, 40, 80, 32, 1001, 161
eng = Engine(L, H, D, C, B, T, U)
torch.cuda.set_stream(eng.stream)
rng = np.random.RandomState(0)
x = torch.as_tensor(rng.randn(T, B, D).astype(np.float32)).cuda()
lengths = torch.full((B,), T, dtype=torch.int32).cuda()
dense = np.zeros((B, U), np.int32)
for b in range(B):
    n = rng.randint(80, 161); dense[b, :n - 1] = rng.randint(1, C - 1, size=n - 1); dense[b, n - 1] = C - 1
dlab = torch.as_tensor(dense).cuda()
def step(ar):
    eng.zero_grads(); eng.mini_batch(x, lengths, dlab, 0.8, 0.5, 1)
    if ar: dist.all_reduce(eng.grads)
    eng.apply(3e-4, 1.0)
for ar in (False, True, False, True):
    for _ in range(3): step(ar)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(10): step(ar)
    torch.cuda.synchronize(); print("all_reduce" if ar else "no all_reduce", (time.time() - t0) / 10 * 1e3, "ms/step")
dist.destroy_process_group()
