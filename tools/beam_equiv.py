"""Dev tool: the host beam search against another build of it (e.g. the previous implementation), on random inputs including
exact ties.  usage: beam_equiv.py old.so new.so"""
import ctypes as C, numpy as np, sys, time
def load(p):
    l = C.CDLL(p); f = l.amdspeech_ctc_beam_search_host
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]; f.restype = C.c_int
    return f
old, new = load(sys.argv[1]), load(sys.argv[2])
def run(f, logits, lengths, w, merge):
    T, B, Cn = logits.shape
    ids = np.zeros((B, T), np.int32); ol = np.zeros(B, np.int32); lp = np.zeros(B, np.float32)
    t0 = time.time()
    rc = f(logits.ctypes.data, lengths.ctypes.data, T, B, Cn, w, merge, ids.ctypes.data, ol.ctypes.data, lp.ctypes.data)
    assert rc == 0
    return ids, ol, lp, time.time() - t0
rng = np.random.RandomState(0)
nbad = ncase = 0
for trial in range(60):
    T, B, Cn = rng.randint(5, 80), rng.randint(1, 6), int(rng.choice([3, 5, 20, 80]))
    scale = rng.choice([0.5, 2.0, 6.0])
    logits = (rng.randn(T, B, Cn) * scale).astype(np.float32)
    if trial % 3 == 0: logits[:, :, Cn - 1] += 4.0
    if trial % 7 == 0: logits = np.round(logits)          # exact ties
    lengths = rng.randint(0, T + 1, size=B).astype(np.int32)
    for w in (1, 3, 25, 100):
        for merge in (0, 1):
            a = run(old, logits, lengths, w, merge); b = run(new, logits, lengths, w, merge); ncase += 1
            if not (np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])):
                nbad += 1
                if nbad < 4: print("MISMATCH", trial, T, B, Cn, w, merge, a[1], b[1], a[2], b[2])
print("mismatches: %d of %d cases" % (nbad, ncase))
logits = (rng.randn(300, 4, 80) * 3).astype(np.float32); lengths = np.full(4, 300, np.int32)
print("first %.2f s   second %.3f s  (T=300, B=4, width 100)" % (run(old, logits, lengths, 100, 1)[3], run(new, logits, lengths, 100, 1)[3]))
