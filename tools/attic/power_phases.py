"""Dev tool: socket power and shader clock (rocm-smi, sampled every 0.2 s) while ONE phase of the step runs in a loop for ~4 s:
   fwd = Engine.forward only, gemm = the weight-gradient products only, step = whole mini-batches, idle."""
import os, subprocess, sys, threading, time, re
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rnn_speech_amd.engine import Engine
from rnn_speech_amd import ops
L, H, D, C, B, T, U = 3, 512, 40, 80, 32, 1001, 161
eng = Engine(L, H, D, C, B, T, U)
rng = np.random.RandomState(0)
x = torch.as_tensor(rng.randn(T, B, D).astype(np.float32)).cuda()
lengths = torch.full((B,), T, dtype=torch.int32).cuda()
dense = np.zeros((B, U), np.int32)
for b in range(B):
    n = 120; dense[b, :n - 1] = rng.randint(1, C - 1, size=n - 1); dense[b, n - 1] = C - 1
dlab = torch.as_tensor(dense).cuda()
a = torch.randn(32032, 1024, device="cuda"); bm = torch.randn(32032, 2048, device="cuda"); out = torch.zeros(1024, 2048, device="cuda")
torch.cuda.set_stream(eng.stream)
samples = []
stop = [False]
def sampler():
    while not stop[0]:
        o = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
        clk = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", o); pw = re.search(r"Power \(W\): ([\d.]+)", o)
        samples.append((int(clk.group(1)) if clk else -1, float(pw.group(1)) if pw else -1))
        time.sleep(0.2)
def phase(name, fn, secs=4.0):
    samples.clear(); stop[0] = False
    th = threading.Thread(target=sampler); th.start()
    t0 = time.time(); n = 0
    while time.time() - t0 < secs:
        for _ in range(10): fn()
        torch.cuda.synchronize(); n += 10
    dt = time.time() - t0
    stop[0] = True; th.join()
    s = samples[len(samples) // 3:]
    print("%-6s %7.3f ms/iter   sclk %4.0f MHz   power %4.0f W   (%d samples)" % (name, dt / n * 1e3, np.mean([c for c, _ in s]), np.mean([p for _, p in s]), len(s)))
def f_fwd(): eng.forward(x, lengths, 0.8, 0.5, 1)
def f_gemm(): ops.gemm(a, bm, trans_a=True, out=out, accumulate=True)
def f_step(): eng.zero_grads(); eng.mini_batch(x, lengths, dlab, 0.8, 0.5, 1); eng.apply(1e-4, 1.0)
def f_idle(): time.sleep(0.01)
for name, fn in (("idle", f_idle), ("fwd", f_fwd), ("gemm", f_gemm), ("step", f_step), ("fwd", f_fwd)):
    phase(name, fn)
