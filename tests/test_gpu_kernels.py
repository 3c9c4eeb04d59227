"""GPU parity tests: every HIP kernel, called through the C ABI (ctypes), against
the numpy oracle on the same seeded inputs.  Tolerances are written per test;
north_star: logits and CTC loss within 1e-3 relative, decoded strings identical."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import model as om  # noqa: E402  (checker only)


def dev(a, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(a)).to(dtype).cuda()


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


@pytest.fixture(scope="module")
def ops():
    from rnn_speech_amd import ops as o
    return o


# ------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K", [(100, 80, 40), (257, 130, 33), (1000, 512, 512), (64, 128, 5000),
                                   (16, 16, 4), (3203, 2048, 64), (512, 2048, 4096),
                                   # the LDS-free kernels' edges: ragged last tiles, K = 33 x 64, a short last band of row tiles
                                   (1000, 200, 2048), (130, 640, 2112), (4100, 512, 2048)])
@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, False), (True, True)])
def test_gemm_matches_fp64(ops, M, N, K, ta, tb):
    rng = np.random.RandomState(M + N + K)
    A = rng.randn(M, K)
    Bm = rng.randn(K, N)
    bias = rng.randn(N)
    a = dev(A.T if ta else A)
    b = dev(Bm.T if tb else Bm)
    out = ops.gemm(a, b, trans_a=ta, trans_b=tb, bias=dev(bias))
    ref = A @ Bm + bias
    assert rel_err(out.cpu().numpy(), ref) < 2e-5
    # accumulate on top of existing contents
    out2 = ops.gemm(a, b, trans_a=ta, trans_b=tb, out=out.clone(), accumulate=True)
    assert rel_err(out2.cpu().numpy(), 2 * ref - bias) < 2e-5


# the dense layers' own shapes (csrc/gemm_skinny.hip): a short N (output Linear, any K incl. a K tail), a short K with the weights
# stored [K,N] (input Linear) or [N,K] (the output layer's input gradient); ragged last row tiles, odd tile counts, every template
@pytest.mark.parametrize("M,N,K,tb", [(32032, 80, 512, False), (1000, 80, 512, False), (300, 96, 200, False), (257, 16, 64, False),
                                      (999, 44, 1028, False), (4099, 64, 68, False), (2000, 28, 2048, False),
                                      (32032, 512, 40, False), (32032, 512, 80, True), (300, 80, 44, False), (300, 80, 44, True),
                                      (1000, 2048, 128, True), (1000, 2048, 120, False), (513, 68, 20, True), (513, 1000, 100, False)])
def test_gemm_short_axis_shapes_match_fp64(ops, M, N, K, tb):
    rng = np.random.RandomState(M + N + K)
    A = rng.randn(M, K)
    Bm = rng.randn(K, N)
    bias = rng.randn(N)
    a = dev(A)
    b = dev(Bm.T if tb else Bm)
    ref = A @ Bm + bias
    out = ops.gemm(a, b, trans_b=tb, bias=dev(bias))
    assert rel_err(out.cpu().numpy(), ref) < 2e-5
    out2 = ops.gemm(a, b, trans_b=tb, out=out.clone(), accumulate=True)
    assert rel_err(out2.cpu().numpy(), 2 * ref - bias) < 2e-5
    # rows and outputs that are views into wider buffers (ld > width), no bias
    wide_a = torch.zeros(M, K + 12, device="cuda")
    wide_a[:, :K] = a
    wide_c = torch.full((M, N + 8), 7.0, device="cuda")
    ops.gemm(wide_a[:, :K], b, trans_b=tb, out=wide_c[:, :N])
    assert rel_err(wide_c[:, :N].cpu().numpy(), A @ Bm) < 2e-5
    assert bool((wide_c[:, N:] == 7.0).all())


@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (100, 80, 40), (257, 130, 33), (1000, 512, 512), (64, 128, 5000),
                                   (768, 4096, 1024), (1024, 4096, 3072), (3203, 1024, 4096), (240, 4096, 1024)])
@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, False), (True, True)])
def test_gemm_bf16x3_matches_fp64(ops, M, N, K, ta, tb):
    """The split-precision GEMM (bf16 hi/lo pairs, hi.hi + hi.lo + lo.hi, f32 accumulate): 16 significant bits per operand, so
    2^-16-ish relative to the product's scale -- ~40x the f32 kernel's error bound, 100x tighter than plain bf16.  Every
    storage layout, ragged tiles, K tails, split K (tall-K shapes) and accumulate."""
    rng = np.random.RandomState(M + N + K)
    if ta and M % 4:            # (a transposed operand's rows are read as they lie: any M; k-contiguous rows want ld % 4 == 0 to vectorise)
        pass
    A = rng.randn(M, K)
    Bm = rng.randn(K, N)
    bias = rng.randn(N)
    a = dev(A.T if ta else A)
    b = dev(Bm.T if tb else Bm)
    out = ops.gemm_bf16x3(a, b, trans_a=ta, trans_b=tb, bias=dev(bias))
    ref = A @ Bm + bias
    scale = np.sqrt(K)          # |sum of K products of unit normals|
    assert np.abs(out.cpu().numpy() - ref).max() < 6e-5 * scale * 4, np.abs(out.cpu().numpy() - ref).max() / scale
    out2 = ops.gemm_bf16x3(a, b, trans_a=ta, trans_b=tb, out=out.clone(), accumulate=True)
    assert np.abs(out2.cpu().numpy() - (2 * ref - bias)).max() < 6e-5 * scale * 8
    # and it IS close to the exact-f32 kernel (same inputs), far closer than a plain bf16 product would be (2^-8)
    exact = ops.gemm(a, b, trans_a=ta, trans_b=tb, bias=dev(bias))
    assert rel_err(out.cpu().numpy(), exact.cpu().numpy()) < 5e-5


# (K a multiple of 32: a k-contiguous operand with a K tail takes the exact-f32 kernel -- a superset in accuracy, not this arithmetic)
@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (257, 130, 64), (1000, 512, 512), (64, 128, 4992), (768, 4096, 1024)])
@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, False), (True, True)])
def test_gemm_bf16_matches_fp64_of_the_rounded_operands(ops, M, N, K, ta, tb):
    """The plain-bf16 GEMM (precision = "bf16": ONE bf16 per value, one MFMA per product, f32 accumulation).  Checked two ways:
    against the float64 product of the ROUNDED operands (the kernel's own arithmetic: only the f32 accumulation is left, 1e-6) and
    against the float64 product of the originals (what the mode costs: 2^-9 per operand, ~3e-3 of the product's scale)."""
    rng = np.random.RandomState(M + 3 * N + K)
    A = rng.randn(M, K).astype(np.float32)
    Bm = rng.randn(K, N).astype(np.float32)
    bias = rng.randn(N).astype(np.float32)
    a = dev(A.T if ta else A)
    b = dev(Bm.T if tb else Bm)
    out = ops.gemm_bf16(a, b, trans_a=ta, trans_b=tb, bias=dev(bias)).cpu().numpy()
    rnd = lambda v: torch.as_tensor(v).to(torch.bfloat16).to(torch.float64).numpy()      # round to nearest even, like the kernel
    scale = np.sqrt(K)
    assert np.abs(out - (rnd(A) @ rnd(Bm) + bias)).max() < 2e-6 * scale * 4
    err = np.abs(out - (A.astype(np.float64) @ Bm.astype(np.float64) + bias)).max() / scale
    assert 1e-4 < err < 2e-2, err          # really bf16 (a split-precision product would sit at 1e-5), and no worse than bf16
    out2 = ops.gemm_bf16(a, b, trans_a=ta, trans_b=tb, out=torch.as_tensor(out).cuda(), accumulate=True).cpu().numpy()
    assert np.abs(out2 - (2 * (rnd(A) @ rnd(Bm)) + bias)).max() < 2e-6 * scale * 8


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (300, 512, 128), (1000, 512, 512), (256, 512, 4096), (2048, 4096, 8192), (3203, 1024, 4096)])
@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, False), (True, True)])
def test_gemm_bf16_packed_matches_fp64_of_the_rounded_operands(ops, M, N, K, ta, tb):
    """The same arithmetic through bf16 COPIES of the operands (round 5, amdspeech_gemm_bf16_packed: a k-contiguous bf16 copy of
    each operand -- a 64 x 64 transpose where the contraction index is the slow one --, a 256 x 256 tile kernel fed by
    global_load_lds, split K through f32 partial tiles).  Ragged M (the clamped rows of the last tile), split K (tall K on few
    tiles), bias, accumulate; shapes the path does not take (a transposed operand that is not a multiple of 64) return None."""
    rng = np.random.RandomState(M + 3 * N + K)
    A = rng.randn(M, K).astype(np.float32)
    Bm = rng.randn(K, N).astype(np.float32)
    bias = rng.randn(N).astype(np.float32)
    a = dev(A.T if ta else A)
    b = dev(Bm.T if tb else Bm)
    out = ops.gemm_bf16_packed(a, b, trans_a=ta, trans_b=tb, bias=dev(bias))
    if ta and M % 64:
        assert out is None
        return
    out = out.cpu().numpy()
    rnd = lambda v: torch.as_tensor(v).to(torch.bfloat16).to(torch.float64).numpy()
    scale = np.sqrt(K)
    assert np.abs(out - (rnd(A) @ rnd(Bm) + bias)).max() < 2e-6 * scale * 4
    # the same values per operand as the converting kernel: the two agree to f32 accumulation order
    other = ops.gemm_bf16(a, b, trans_a=ta, trans_b=tb, bias=dev(bias)).cpu().numpy() if K % 32 == 0 else None
    if other is not None:
        assert np.abs(out - other).max() < 2e-6 * scale * 4
    out2 = ops.gemm_bf16_packed(a, b, trans_a=ta, trans_b=tb, out=torch.as_tensor(out).cuda(), accumulate=True).cpu().numpy()
    assert np.abs(out2 - (2 * (rnd(A) @ rnd(Bm)) + bias)).max() < 2e-6 * scale * 8


def test_linear_bwd(ops):
    rng = np.random.RandomState(5)
    M, K, N = 777, 40, 96
    x, w, dy = rng.randn(M, K), rng.randn(K, N), rng.randn(M, N)
    dw0, db0 = rng.randn(K, N), rng.randn(N)
    dw, db = dev(dw0), dev(db0)
    dx = ops.linear_bwd(dev(x), dev(w), dev(dy), dw, db, need_dx=True)
    assert rel_err(dx.cpu().numpy(), dy @ w.T) < 2e-5
    assert rel_err(dw.cpu().numpy(), dw0 + x.T @ dy) < 2e-5
    assert rel_err(db.cpu().numpy(), db0 + dy.sum(0)) < 2e-5


# the dense layers' weight / bias gradients at their own shapes (csrc/gemm_skinny.hip, the 32k-row reduction onto a narrow output):
# small operand = x (input layer: [rows, 40] / [rows, 120]) or = dy (output layer: [rows, 80]); ragged row counts, a last row chunk
# that is mostly empty, widths that are not multiples of 64, the appended ones-column landing in a second fragment (K_in = 64)
@pytest.mark.parametrize("M,K,N", [(32032, 40, 512), (32032, 512, 80), (4099, 120, 1024), (5000, 1024, 80), (4100, 64, 132),
                                   (4100, 132, 64), (9001, 124, 200), (6000, 300, 4)])
def test_linear_bwd_dense_layer_shapes(ops, M, K, N):
    rng = np.random.RandomState(M + K + N)
    x, w, dy = rng.randn(M, K), rng.randn(K, N), rng.randn(M, N)
    dw0, db0 = rng.randn(K, N), rng.randn(N)
    dw, db = dev(dw0), dev(db0)
    dx = ops.linear_bwd(dev(x), dev(w), dev(dy), dw, db, need_dx=True)
    assert rel_err(dx.cpu().numpy(), dy @ w.T) < 2e-5
    assert rel_err(dw.cpu().numpy(), dw0 + x.T @ dy) < 2e-5
    assert rel_err(db.cpu().numpy(), db0 + dy.sum(0)) < 2e-5
    # the plain product (no accumulate, no column sums) through the same kernel
    out = ops.gemm(dev(x), dev(dy), trans_a=True)
    assert rel_err(out.cpu().numpy(), x.T @ dy) < 2e-5


# ------------------------------------------------------------------------- CTC
def make_ctc_case(T, B, C, U, seed, lengths=None):
    rng = np.random.RandomState(seed)
    logits = rng.randn(T, B, C).astype(np.float32) * 2.0
    if lengths is None:
        lengths = rng.randint(max(1, T // 2), T + 1, size=B)
    dense = np.zeros((B, U), np.int32)
    for b in range(B):
        n = rng.randint(1, max(2, min(U - 1, int(lengths[b]) // 2 + 1)))
        lab = rng.randint(1, C - 1, size=n)
        if n > 2 and b % 2 == 0:
            lab[1] = lab[0]                 # repeated label -> mandatory blank between
        dense[b, :n] = lab
        dense[b, n] = C - 1                 # EOS (doubles as blank)
    return logits, dense, np.asarray(lengths, np.int32)


@pytest.mark.parametrize("T,B,C,U", [(30, 4, 80, 12), (101, 7, 80, 40), (257, 3, 80, 161), (64, 2, 29, 70),
                                      (300, 2, 80, 600), (600, 2, 80, 1100)])
def test_ctc_loss_and_grad(ops, T, B, C, U):
    logits, dense, lengths = make_ctc_case(T, B, C, U, seed=T + B)
    loss, dl = ops.ctc_loss_fwd_bwd(dev(logits), dev(dense, torch.int32), dev(lengths, torch.int32))
    rows = om.sparsify_labels(dense, C)
    ref_loss, ref_dl = om.ctc_loss_and_grad(logits.astype(np.float64), rows, lengths)
    assert np.all(ref_loss > 0)
    np.testing.assert_allclose(loss.cpu().numpy(), ref_loss, rtol=1e-3)      # north_star tolerance
    assert rel_err(loss.cpu().numpy(), ref_loss) < 2e-5
    assert np.abs(dl.cpu().numpy() - ref_dl).max() < 2e-3


@pytest.mark.parametrize("env", [{"AMDSPEECH_CTC_SHIFT": "0"}, {"AMDSPEECH_CTC_SHIFT": "0", "AMDSPEECH_CTC_PAIR": "0"}],
                         ids=["two-frames-per-lds-exchange", "one-frame-per-lds-exchange"])
def test_ctc_fallback_recursion_kernels_keep_parity(env):
    """AMDSPEECH_CTC_SHIFT=0 takes the DPP-shift recursion kernel (the default for 129..384 extended states) out and leaves the
    LDS-exchange kernels: two frames per exchange, or -- with AMDSPEECH_CTC_PAIR=0 -- one.  The library reads the switches once
    per process, so the oracle cases run again in a child."""
    import os
    import subprocess
    import sys
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-q", "-x", "-k", "test_ctc_loss_and_grad"],
                         env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]


def test_ctc_reference_label_conventions(ops):
    """Label id 0 dropped, EOS ends the target but counts in required_time, empty row ->
    [C-1], zero-length / too-short rows -> loss 0 and gradient 0."""
    T, C, U = 20, 80, 10
    rng = np.random.RandomState(3)
    dense = np.zeros((6, U), np.int32)
    dense[0, :5] = [5, 0, 6, 7, 79]           # the 0 disappears
    dense[1, :4] = [5, 6, 7, 79]              # same target as row 0 -> same loss
    dense[2, :] = 0                            # empty -> [79]: all-blank path
    dense[3, :6] = [1, 2, 3, 4, 5, 79]        # required 6 > len 5 -> ignored
    dense[4, :3] = [9, 9, 79]                 # len 0 -> ignored
    dense[5, :5] = [9, 79, 11, 0, 0]          # label after EOS: target stops at EOS
    lengths = np.array([20, 20, 15, 5, 0, 12], np.int32)
    logits = rng.randn(T, 6, C).astype(np.float32)
    logits[:, 1] = logits[:, 0]
    loss, dl = ops.ctc_loss_fwd_bwd(dev(logits), dev(dense, torch.int32), dev(lengths, torch.int32))
    loss, dl = loss.cpu().numpy(), dl.cpu().numpy()
    ref_loss, ref_dl = om.ctc_loss_and_grad(logits.astype(np.float64), om.sparsify_labels(dense, C), lengths)
    np.testing.assert_allclose(loss, ref_loss, rtol=1e-4, atol=1e-6)
    assert abs(loss[0] - loss[1]) < 1e-4 * loss[0]
    assert loss[3] == 0 and loss[4] == 0 and not dl[:, 3].any() and not dl[:, 4].any()
    assert not dl[15:, 2].any()
    assert np.abs(dl - ref_dl).max() < 1e-3


def test_ctc_impossible_alignment(ops):
    """Repeats need more frames than required_time checks: p = 0 -> loss inf, grad = softmax."""
    T, C, U = 4, 10, 6
    dense = np.zeros((1, U), np.int32)
    dense[0, :4] = [3, 3, 3, 9]
    lengths = np.array([4], np.int32)
    logits = np.random.RandomState(0).randn(T, 1, C).astype(np.float32)
    loss, dl = ops.ctc_loss_fwd_bwd(dev(logits), dev(dense, torch.int32), dev(lengths, torch.int32))
    assert np.isinf(loss.cpu().numpy()[0])
    y = np.exp(logits - logits.max(2, keepdims=True))
    y /= y.sum(2, keepdims=True)
    assert np.abs(dl.cpu().numpy() - y).max() < 1e-5


def test_greedy_decode(ops):
    T, B, C = 300, 5, 80
    rng = np.random.RandomState(11)
    logits = rng.randn(T, B, C).astype(np.float32)
    logits[:, :, C - 1] += 1.5                       # plenty of blanks
    logits[10:20, 0, 7] += 10                       # a long run that must collapse
    lengths = np.array([300, 257, 1, 0, 64], np.int32)
    ids, out_len = ops.ctc_greedy_decode(dev(logits), dev(lengths, torch.int32))
    ref = om.greedy_decode(logits, lengths)
    ids, out_len = ids.cpu().numpy(), out_len.cpu().numpy()
    for b in range(B):
        assert out_len[b] == len(ref[b])
        assert list(ids[b, :out_len[b]]) == ref[b]
        assert np.all(ids[b, out_len[b]:] == C)      # reference pads with num_labels (:718)


# ------------------------------------------------------------------- optimiser
@pytest.mark.parametrize("n,clip", [(1000, 1.0), (6359632 // 8 + 3, 1.0), (4096, 1e9)])
def test_clip_adam(ops, n, clip):
    rng = np.random.RandomState(n % 1000)
    p = {"w": rng.randn(n).astype(np.float32)}
    g = {"w": (rng.randn(n) * 0.1).astype(np.float32)}
    m = {"w": np.zeros(n, np.float32)}
    v = {"w": np.zeros(n, np.float32)}
    dp, dg, dm, dv = dev(p["w"]), dev(g["w"]), dev(m["w"]), dev(v["w"])
    import math
    for step in (1, 2, 3):
        lr = 3e-4
        gn = om.clip_and_adam(p, g, m, v, step, lr, clip)
        lr_t = lr * math.sqrt(1 - 0.999 ** step) / (1 - 0.9 ** step)
        norm = ops.clip_adam(dp, dg, dm, dv, clip, lr_t)
        assert abs(float(norm.cpu()) - gn) < 1e-4 * gn
        assert np.abs(dp.cpu().numpy() - p["w"]).max() < 2e-6
        assert rel_err(dm.cpu().numpy(), m["w"]) < 1e-4
        assert rel_err(dv.cpu().numpy(), v["w"]) < 1e-4


def test_edit_distance_and_merge_repeated_on_device(ops):
    rng = np.random.RandomState(3)
    n, la, lb = 40, 150, 70
    a = rng.randint(0, 6, size=(n, la)).astype(np.int32)
    b = rng.randint(0, 6, size=(n, lb)).astype(np.int32)
    alen = rng.randint(0, la + 1, size=n).astype(np.int32)
    blen = rng.randint(0, lb + 1, size=n).astype(np.int32)
    alen[0], blen[0] = 0, 0
    alen[1], blen[1] = la, lb
    alen[2] = 0
    b[3, :64] = a[3, :64]; alen[3] = blen[3] = 64          # identical -> 0
    out = ops.edit_distance(dev(a, torch.int32), dev(alen, torch.int32), dev(b, torch.int32), dev(blen, torch.int32))
    out = out.cpu().numpy()
    for i in range(n):
        assert out[i] == om.edit_distance(a[i, :alen[i]], b[i, :blen[i]]), i
    assert out[3] == 0
    ids = rng.randint(0, 3, size=(5, 300)).astype(np.int32)
    lens = np.array([300, 1, 0, 257, 64], np.int32)
    d_ids, d_len = dev(ids, torch.int32), dev(lens, torch.int32)
    ops.merge_repeated(d_ids, d_len, 99)
    got, gl = d_ids.cpu().numpy(), d_len.cpu().numpy()
    for r in range(5):
        row = ids[r, :lens[r]]
        ref = [int(v) for i, v in enumerate(row) if i == 0 or v != row[i - 1]]
        assert gl[r] == len(ref) and list(got[r, :gl[r]]) == ref
        assert np.all(got[r, gl[r]:lens[r]] == 99)


def test_ctc_staged_call_equals_the_single_call(ops):
    """amdspeech_ctc_loss_fwd_bwd_staged: stage 1 (targets + log-softmax) then stage 2 (recursions + gradient), with other
    work in between, gives bit-identical loss and gradient to the single call."""
    logits, dense, lengths = make_ctc_case(60, 5, 80, 14, seed=3)
    lg, dl, ln = dev(logits, torch.float32), torch.as_tensor(dense).cuda(), torch.as_tensor(lengths).cuda()
    loss0, d0 = ops.ctc_loss_fwd_bwd(lg, dl, ln)
    ws = ops.CtcWorkspace(60, 5, 80, 14, lg.device)
    loss1, d1 = torch.empty_like(loss0), torch.empty_like(d0)
    ops.ctc_loss_fwd_bwd(lg, dl, ln, ws=ws, loss=loss1, dlogits=d1, stage=1)
    torch.cuda.synchronize()
    ops.ctc_loss_fwd_bwd(lg, dl, ln, ws=ws, loss=loss1, dlogits=d1, stage=2)
    assert torch.equal(loss0, loss1) and torch.equal(d0, d1)
