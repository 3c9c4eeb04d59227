"""Dev tool: HIP-event times of the two recurrence kernels (amdspeech_profile_*) at a config, per time step."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rnn_speech_amd.engine import Engine
from rnn_speech_amd import lib as _lib
nums = [int(v) for v in sys.argv[1:] if not v.startswith('--')]
L, H, D, C, B, T, U = nums if len(nums) == 7 else (3, 512, 40, 80, 32, 1001, 161)
eng = Engine(L, H, D, C, B, T, U, precision='bf16x3' if '--bf16x3' in sys.argv else 'f32')
rng = np.random.RandomState(0)
x = torch.as_tensor(rng.randn(T, B, D).astype(np.float32)).cuda()
lengths = torch.full((B,), T, dtype=torch.int32).cuda()
dense = np.zeros((B, U), np.int32)
for b in range(B):
    n = min(U, max(2, T // 8)); dense[b, :n - 1] = rng.randint(1, C - 1, size=n - 1); dense[b, n - 1] = C - 1
dlab = torch.as_tensor(dense).cuda()
lib = _lib.load(); lib.amdspeech_profile_enable(1)
torch.cuda.set_stream(eng.stream)
fw, bw = [], []
for i in range(14):
    eng.zero_grads(); eng.mini_batch(x, lengths, dlab, 0.8, 0.5, i + 1); torch.cuda.synchronize()
    ms, nl = ctypes.c_float(), ctypes.c_int()
    lib.amdspeech_profile_get(0, ctypes.byref(ms), ctypes.byref(nl)); fw.append(ms.value)
    lib.amdspeech_profile_get(1, ctypes.byref(ms), ctypes.byref(nl)); bw.append(ms.value)
eng.check()
print("grad abs max %.4g finite %s" % (float(eng.grads.abs().max()), bool(torch.isfinite(eng.grads).all())))
print("bwd ms per call:", " ".join("%.3f" % v for v in bw), " fwd:", " ".join("%.3f" % v for v in fw))
print("fwd ms %.3f (%.2f us/step)  bwd ms %.3f (%.2f us/step)  steps %d  loss %.4f" % (
    min(fw[1:]), min(fw[1:]) * 1e3 / nl.value, min(bw[1:]), min(bw[1:]) * 1e3 / nl.value, nl.value, float(eng.loss.mean())))
