"""Dev tool: amdspeech_gemm_bf16_packed against torch on bf16-rounded operands, and its rate at the products of configs[4]."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rnn_speech_amd import lib as L
lib = L.load()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)
S = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)

def run(ta, tb, M, N, K, bias=False, acc=False, check=True, reps=5):
    torch.manual_seed(0)
    A = (torch.randn((K, M) if ta else (M, K), device="cuda") * 0.5)
    B = (torch.randn((N, K) if tb else (K, N), device="cuda") * 0.5)
    bv = torch.randn(N, device="cuda") if bias else None
    C0 = torch.randn(M, N, device="cuda")
    Cc = C0.clone()
    n = lib.amdspeech_gemm_bf16_packed_scratch_bytes(int(ta), int(tb), M, N, K, A.shape[1], B.shape[1])
    assert n > 0, "shape not taken"
    scr = torch.empty(n, dtype=torch.uint8, device="cuda")
    def call(out):
        L.check(lib.amdspeech_gemm_bf16_packed(S(), int(ta), int(tb), M, N, K, P(A), A.shape[1], P(B), B.shape[1], P(out), N, P(bv), int(acc),
                                               P(scr), n), "gemm_bf16_packed")
    call(Cc)
    torch.cuda.synchronize()
    err = None
    if check:
        Ar, Br = A.bfloat16().float(), B.bfloat16().float()
        ref = (Ar.t() if ta else Ar).double() @ (Br.t() if tb else Br).double()
        if bias: ref = ref + bv.double()
        if acc: ref = ref + C0.double()
        err = float((Cc.double() - ref).abs().max() / ref.abs().max())
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    out = torch.empty(M, N, device="cuda")
    call(out); torch.cuda.synchronize()
    t0.record()
    for _ in range(reps): call(out)
    t1.record(); torch.cuda.synchronize()
    ms = t0.elapsed_time(t1) / reps
    print("ta=%d tb=%d M=%d N=%d K=%d bias=%d acc=%d: rel err %s, %.3f ms incl. the copies = %.0f TFLOP/s" % (ta, tb, M, N, K, bias, acc, err, ms, 2.0 * M * N * K / ms / 1e9))

if __name__ == "__main__":
    run(0, 0, 512, 512, 256, bias=True)
    run(0, 1, 300, 256, 128, acc=True)
    run(1, 0, 256, 512, 4096)              # split K
    run(1, 0, 2048, 4096, 8192, acc=True, reps=3)
    TB = 63872
    run(0, 0, TB, 4096, 1024, bias=True, check=False)     # x . W_ih
    run(0, 1, TB, 1024, 4096, check=False)                # dX
    run(1, 0, 2048, 4096, TB, acc=True, check=False)      # dK, both halves stacked
