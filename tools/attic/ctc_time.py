"""Dev tool: the CTC stage at the benchmark shape (T = 1001, B = 32, C = 160, 161-label targets), per-call time of stage 2
(alpha/beta + gradient) and of the whole call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rnn_speech_amd import ops
T, B, C, U = 1001, 32, 160, 161
g = torch.Generator().manual_seed(0)
logits = torch.randn(T, B, C, generator=g).cuda()
labels = torch.randint(1, C - 1, (B, U), generator=g, dtype=torch.int32).cuda()
lengths = torch.full((B,), T, dtype=torch.int32).cuda()
ws = ops.CtcWorkspace(T, B, C, U, logits.device)
loss_buf = torch.empty(B, device="cuda"); dl = torch.empty_like(logits)
def run(stage):
    return ops.ctc_loss_fwd_bwd(logits, labels, lengths, ws=ws, loss=loss_buf, dlogits=dl, stage=stage)
out = run(0); torch.cuda.synchronize()
loss = out[0] if isinstance(out, (tuple, list)) else out
print("loss sum %.6f" % float(loss.sum()))
for stage in (0, 2):
    for _ in range(3): run(stage)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run(stage)
    e1.record(); torch.cuda.synchronize()
    print("stage %d: %.1f us per call" % (stage, e0.elapsed_time(e1) * 1e3 / 20))
