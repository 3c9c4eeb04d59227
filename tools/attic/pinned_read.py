"""Dev tool: is pinned host memory slow to read from the CPU on this box?  (the asynchronous decoder reads its logits from it)"""
import time, numpy as np, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rnn_speech_amd import ops
T, B, C = 1001, 32, 80
x = torch.randn(T, B, C)
p = torch.empty(T, B, C).pin_memory(); p.copy_(x)
for name, t in (("pageable", x), ("pinned", p)):
    a = t.numpy()
    t0 = time.perf_counter(); b = a.copy(); t1 = time.perf_counter(); s = float(a.sum()); t2 = time.perf_counter()
    lens = np.full(B, T, np.int32)
    ops.ctc_beam_search(a, lens, 100, True)
    t3 = time.perf_counter(); ops.ctc_beam_search(a, lens, 100, True); t4 = time.perf_counter()
    ops.ctc_beam_search(a, lens, 100, True, max_threads=16); t5 = time.perf_counter()
    print("%-9s copy of 10 MB %.2f ms, sum %.2f ms, beam search of the batch %.1f ms (32 threads), %.1f ms (16 threads)"
          % (name, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t4 - t3) * 1e3, (t5 - t4) * 1e3))
