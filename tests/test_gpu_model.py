"""GPU parity of the whole hot path (Linear -> stacked LSTM -> Linear -> CTC ->
BPTT -> clip + Adam) against the numpy oracle, through the C ABI."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import model as om  # noqa: E402  (checker only)


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def make_batch(T, B, D, C, U, seed, full=False):
    rng = np.random.RandomState(seed)
    x = rng.randn(T, B, D).astype(np.float32)
    lengths = np.full(B, T, np.int32) if full else rng.randint(min(max(2, T // 2), T), T + 1, size=B).astype(np.int32)
    if B > 2 and not full:
        lengths[1] = 0                       # padded row of a short final batch
    dense = np.zeros((B, U), np.int32)
    for b in range(B):
        n = rng.randint(1, max(2, min(U - 1, int(lengths[b]) // 3 + 1)))
        dense[b, :n] = rng.randint(1, C - 1, size=n)
        dense[b, n] = C - 1
    return x, lengths, dense


CONFIGS = [
    # L, H, D, C, B, T, U
    (1, 16, 8, 80, 2, 12, 6),
    (2, 32, 20, 80, 5, 25, 10),
    (3, 64, 40, 80, 33, 40, 16),       # B > 32: two batch blocks, ragged
    (1, 128, 40, 80, 2, 101, 40),      # cfg1 shape (plumbing config), shortened in time
    (2, 48, 120, 80, 10, 30, 12),      # H = 48 (not a power of two), reference default batch 10
    (5, 1024, 120, 80, 64, 12, 6),     # BASELINE configs[2] shape (5x1024, 120-dim, batch 64), short in time
    (3, 512, 40, 80, 32, 16, 8),       # BASELINE configs[1] shape, short in time
    (3, 1024, 120, 80, 10, 10, 5),     # the reference's pre-trained model shape (3x1024 fbank, batch 10)
    (2, 256, 40, 80, 20, 24, 8),       # dataflow kernels: H = 256, ragged second batch tile (B = 20)
    (4, 384, 40, 80, 9, 18, 6),        # dataflow kernels: H = 384, 4 layers, one batch tile
    (2, 256, 40, 80, 64, 14, 6),       # dataflow kernels: 2 layers x 4 batch tiles = all 8 XCDs carry a group
    (1, 128, 20, 80, 100, 70, 20),     # dataflow kernels: 7 batch tiles of one layer; T >= 64 -> in-kernel GEMM workers
    # round 4: a layer of the backward dataflow kernel trails the one above by ~9 steps and drains for 9 more -- sequences SHORTER than
    # that lag (no steady-state trip at all), incl. a single frame, on 3 layers x 2 batch tiles and on the per-layer H = 1024 kernels
    (3, 128, 20, 80, 20, 1, 2),
    (3, 128, 20, 80, 20, 2, 2),
    (3, 256, 20, 80, 20, 5, 3),
    (3, 512, 40, 80, 32, 9, 4),
    (2, 1024, 40, 80, 20, 2, 2),
]


@pytest.mark.parametrize("L,H,D,C,B,T,U", CONFIGS)
def test_forward_backward_adam_parity(L, H, D, C, B, T, U):
    from rnn_speech_amd.engine import Engine
    eng = Engine(L, H, D, C, B, T, U, seed=7)
    rng = np.random.RandomState(1)
    p = eng.to_numpy()
    for k in p:                               # non-zero biases exercise the bias paths
        if p[k].ndim == 1:
            p[k] = (rng.randn(*p[k].shape) * 0.1).astype(np.float32)
    eng.load_numpy(p)
    x, lengths, dense = make_batch(T, B, D, C, U, seed=L * 100 + H)
    dx = torch.as_tensor(x).cuda()
    dlen = torch.as_tensor(lengths).cuda()
    dlab = torch.as_tensor(dense).cuda()

    p64 = {k: v.astype(np.float64) for k, v in p.items()}
    logits_ref, final_ref, cache = om.forward(p64, x.astype(np.float64), lengths, L, keep_cache=True)
    rows = om.sparsify_labels(dense, C)
    loss_ref, dl_ref = om.ctc_loss_and_grad(logits_ref, rows, lengths)
    g_ref = om.backward(p64, cache, dl_ref, lengths, L)

    eng.zero_grads()
    eng.mini_batch(dx, dlen, dlab)
    torch.cuda.synchronize()
    logits = eng.logits.cpu().numpy()
    # north_star: logits and CTC loss within 1e-3 relative
    assert rel_err(logits, logits_ref) < 1e-4
    loss = eng.loss.cpu().numpy()
    np.testing.assert_allclose(loss, loss_ref, rtol=1e-3, atol=1e-5)
    h, c = eng.final_state()
    for l in range(L):
        assert np.abs(c[l].cpu().numpy() - final_ref[l][0]).max() < 1e-4
        assert np.abs(h[l].cpu().numpy() - final_ref[l][1]).max() < 1e-4
    g = eng.to_numpy(eng.grads)
    for k in g_ref:
        assert rel_err(g[k], g_ref[k]) < 2e-3, k
    # greedy-decoded label strings identical
    from rnn_speech_amd import ops
    ids, out_len = ops.ctc_greedy_decode(eng.logits, dlen)
    ref_ids = om.greedy_decode(logits_ref, lengths)
    ids, out_len = ids.cpu().numpy(), out_len.cpu().numpy()
    for b in range(B):
        assert list(ids[b, :out_len[b]]) == ref_ids[b]

    # second mini-batch accumulates (reference accumulators, :391-401), then clip + Adam
    x2, len2, dense2 = make_batch(T, B, D, C, U, seed=999)
    eng.mini_batch(torch.as_tensor(x2).cuda(), torch.as_tensor(len2).cuda(), torch.as_tensor(dense2).cuda())
    lg2, _, cache2 = om.forward(p64, x2.astype(np.float64), len2, L, keep_cache=True)
    _, dl2 = om.ctc_loss_and_grad(lg2, om.sparsify_labels(dense2, C), len2)
    g2 = om.backward(p64, cache2, dl2, len2, L)
    acc = {k: g_ref[k] + g2[k] for k in g_ref}
    g = eng.to_numpy(eng.grads)
    for k in acc:
        assert rel_err(g[k], acc[k]) < 2e-3, k
    m = {k: np.zeros_like(v) for k, v in p64.items()}
    v = {k: np.zeros_like(vv) for k, vv in p64.items()}
    pn = {k: vv.copy() for k, vv in p64.items()}
    gn = om.clip_and_adam(pn, acc, m, v, 1, 3e-4, 1.0)
    norm = eng.apply(3e-4, 1.0)
    assert abs(float(norm.cpu()) - gn) <= 2e-3 * gn + 1e-12      # (T = 1: no feasible alignment, every gradient is exactly 0)
    pd = eng.to_numpy()
    for k in pn:
        # first Adam step moves every weight by ~+-lr; compare the update, not the weight, and only
        # where the gradient sign is numerically determined (a ~0 gradient flips between f32 and f64)
        sure = np.abs(acc[k]) > 1e-3 * np.abs(acc[k]).max()
        if sure.any():
            assert np.abs((pd[k] - p[k]) - (pn[k] - p64[k]))[sure].max() < 0.05 * 3e-4, k
        else:                                   # (an all-zero gradient: nothing moves)
            assert np.abs(pd[k] - p[k]).max() < 1e-9, k


BOTH_PATHS = pytest.mark.parametrize("H", [64, 128], ids=["step-kernels", "dataflow-kernels"])


@BOTH_PATHS
@pytest.mark.parametrize("keep", [(1.0, 1.0), (0.8, 0.5)])
def test_short_batch_stops_at_longest_utterance(keep, H):
    """dynamic_rnn semantics (reference :276-278): a batch whose longest utterance is shorter than the
    padded length runs only that many frames.  Same logits / loss / gradients / final state as the full-
    length run (and as the oracle), and a following full-length batch sees no stale data."""
    from rnn_speech_amd.engine import Engine
    L, D, C, B, T, U = 3, 20, 80, 6, 40, 10
    x, lengths, dense = make_batch(T, B, D, C, U, seed=5)
    lengths = np.minimum(lengths, 23).astype(np.int32)
    lengths[2] = 23
    dx, dlen, dlab = torch.as_tensor(x).cuda(), torch.as_tensor(lengths).cuda(), torch.as_tensor(dense).cuda()
    outs = []
    for max_len in (None, 23, 1000):
        eng = Engine(L, H, D, C, B, T, U, seed=11)
        eng.logits.fill_(123.0)
        eng.dlogits.fill_(123.0)                       # stale garbage must not survive in the tail
        eng.zero_grads()
        eng.mini_batch(dx, dlen, dlab, keep[0], keep[1], seed=9, max_len=max_len)
        h, c = eng.final_state()
        outs.append(dict(logits=eng.logits.cpu().numpy().copy(), dlogits=eng.dlogits.cpu().numpy().copy(),
                         loss=eng.loss.cpu().numpy().copy(), g=eng.grads.cpu().numpy().copy(),
                         h=h.cpu().numpy().copy(), c=c.cpu().numpy().copy(), eng=eng))
    full, short, clamped = outs
    assert short["eng"]._Tr == 23 and full["eng"]._Tr == T and clamped["eng"]._Tr == T
    for k in ("logits", "dlogits", "loss", "h", "c"):
        np.testing.assert_allclose(short[k], full[k], rtol=0, atol=1e-6, err_msg=k)
    assert np.abs(short["g"] - full["g"]).max() < 1e-5 * np.abs(full["g"]).max()   # split-K order differs
    if keep == (1.0, 1.0):
        p64 = {k: v.astype(np.float64) for k, v in short["eng"].to_numpy().items()}
        ref, _, _ = om.forward(p64, x.astype(np.float64), lengths, L)
        assert rel_err(short["logits"], ref) < 1e-4
    # a full-length batch on the engine that just ran the short one
    eng = short["eng"]
    x2, len2, dense2 = make_batch(T, B, D, C, U, seed=6, full=True)
    eng.zero_grads()
    eng.mini_batch(torch.as_tensor(x2).cuda(), torch.as_tensor(len2).cuda(), torch.as_tensor(dense2).cuda(),
                   max_len=int(len2.max()))
    fresh = Engine(L, H, D, C, B, T, U, seed=11)
    fresh.zero_grads()
    fresh.mini_batch(torch.as_tensor(x2).cuda(), torch.as_tensor(len2).cuda(), torch.as_tensor(dense2).cuda())
    np.testing.assert_allclose(eng.logits.cpu().numpy(), fresh.logits.cpu().numpy(), atol=1e-6)
    assert np.abs((eng.grads - fresh.grads).cpu().numpy()).max() < 1e-5 * float(fresh.grads.abs().max())


@BOTH_PATHS
def test_state_carry_and_reset(H):
    """Persistent RNN state across mini-batches (reference :266-298)."""
    from rnn_speech_amd.engine import Engine
    L, D, C, B, T, U = 2, 8, 80, 3, 10, 5
    eng = Engine(L, H, D, C, B, T, U, seed=3)
    p64 = {k: v.astype(np.float64) for k, v in eng.to_numpy().items()}
    x1, len1, _ = make_batch(T, B, D, C, U, seed=1, full=True)
    x2, len2, _ = make_batch(T, B, D, C, U, seed=2, full=True)
    _, st, _ = om.forward(p64, x1.astype(np.float64), len1, L)
    ref2, _, _ = om.forward(p64, x2.astype(np.float64), len2, L, state=st)
    eng.forward(torch.as_tensor(x1).cuda(), torch.as_tensor(len1).cuda(), use_state=True)
    eng.keep_state()
    out = eng.forward(torch.as_tensor(x2).cuda(), torch.as_tensor(len2).cuda(), use_state=True)
    assert rel_err(out.cpu().numpy(), ref2) < 1e-4
    eng.zero_state()
    ref0, _, _ = om.forward(p64, x2.astype(np.float64), len2, L)
    out = eng.forward(torch.as_tensor(x2).cuda(), torch.as_tensor(len2).cuda(), use_state=True)
    assert rel_err(out.cpu().numpy(), ref0) < 1e-4


@BOTH_PATHS
def test_dropout_is_consistent_between_forward_and_backward(H):
    """With keep < 1 the masks are a pure function of (seed, layer, element): same seed ->
    same logits; gradients match a directional finite difference of the SAME masked net."""
    from rnn_speech_amd.engine import Engine
    L, D, C, B, T, U = 2, 8, 80, 4, 16, 6
    eng = Engine(L, H, D, C, B, T, U, seed=5)
    x, lengths, dense = make_batch(T, B, D, C, U, seed=4, full=True)
    dx, dlen, dlab = torch.as_tensor(x).cuda(), torch.as_tensor(lengths).cuda(), torch.as_tensor(dense).cuda()
    a = eng.forward(dx, dlen, 0.8, 0.5, seed=42).clone()
    b = eng.forward(dx, dlen, 0.8, 0.5, seed=42).clone()
    c = eng.forward(dx, dlen, 0.8, 0.5, seed=43).clone()
    assert torch.equal(a, b) and not torch.equal(a, c)
    ztop = eng.lstm_ws.ztop
    frac_zero = float((ztop == 0).float().mean().cpu())
    assert 0.4 < frac_zero < 0.6                       # output keep 0.5
    eng.zero_grads()
    eng.mini_batch(dx, dlen, dlab, 0.8, 0.5, seed=42)
    g = eng.grads.clone()
    base = float(eng.loss.sum().cpu())
    direction = torch.randn_like(eng.params) * (eng.params != 0).float()
    direction /= direction.norm()
    eps = 1e-2
    saved = eng.params.clone()
    eng.params.add_(direction, alpha=eps)
    eng.mini_batch(dx, dlen, dlab, 0.8, 0.5, seed=42, compute_gradients=False)
    plus = float(eng.loss.sum().cpu())
    eng.params.copy_(saved).add_(direction, alpha=-eps)
    eng.mini_batch(dx, dlen, dlab, 0.8, 0.5, seed=42, compute_gradients=False)
    minus = float(eng.loss.sum().cpu())
    fd = (plus - minus) / (2 * eps)
    an = float((g * direction).sum().cpu())
    assert abs(fd - an) < 0.05 * max(1.0, abs(an)), (fd, an, base)


@BOTH_PATHS
def test_batch_normalization_option(H):
    """config `batch_normalization : True` (reference :253-259): moments over the batch axis per
    (t, feature), eps 1e-3, no affine; forward + backward parity with the oracle."""
    from rnn_speech_amd.engine import Engine
    L, D, C, B, T, U = 2, 12, 80, 6, 18, 7
    eng = Engine(L, H, D, C, B, T, U, seed=21, normalization=True)
    x, lengths, dense = make_batch(T, B, D, C, U, seed=77)
    p64 = {k: v.astype(np.float64) for k, v in eng.to_numpy().items()}
    logits_ref, _, cache = om.forward(p64, x.astype(np.float64), lengths, L, keep_cache=True, normalization=True)
    loss_ref, dl_ref = om.ctc_loss_and_grad(logits_ref, om.sparsify_labels(dense, C), lengths)
    g_ref = om.backward(p64, cache, dl_ref, lengths, L)
    eng.zero_grads()
    eng.mini_batch(torch.as_tensor(x).cuda(), torch.as_tensor(lengths).cuda(), torch.as_tensor(dense).cuda())
    assert rel_err(eng.logits.cpu().numpy(), logits_ref) < 1e-4
    np.testing.assert_allclose(eng.loss.cpu().numpy(), loss_ref, rtol=1e-3, atol=1e-5)
    g = eng.to_numpy(eng.grads)
    for k in g_ref:
        if k == "input_b":        # a bias in front of a batch norm has exactly zero gradient
            assert np.abs(g[k]).max() < 1e-4 * np.abs(g["input_w"]).max()
            continue
        assert rel_err(g[k], g_ref[k]) < 2e-3, k


@pytest.mark.parametrize("L,H,D,C,B,T,U", [(2, 32, 20, 80, 5, 25, 10), (3, 64, 40, 80, 33, 40, 16),
                                            (1, 128, 40, 80, 2, 101, 40), (3, 512, 40, 80, 32, 16, 8),
                                            (5, 1024, 120, 80, 64, 12, 6), (2, 256, 40, 80, 20, 40, 12)])
def test_bf16x3_option_parity(L, H, D, C, B, T, U):
    """precision='bf16x3' (opt-in): products as hi.hi + hi.lo + lo.hi on bf16 MFMA, f32 accumulate.
    Operands keep 16 significant bits, so the tolerances are those of the f32 path times ~50 -- still
    far inside north_star's 1e-3 on logits and loss."""
    from rnn_speech_amd.engine import Engine
    eng = Engine(L, H, D, C, B, T, U, seed=7, precision="bf16x3")
    x, lengths, dense = make_batch(T, B, D, C, U, seed=L * 100 + H)
    p64 = {k: v.astype(np.float64) for k, v in eng.to_numpy().items()}
    logits_ref, final_ref, cache = om.forward(p64, x.astype(np.float64), lengths, L, keep_cache=True)
    loss_ref, dl_ref = om.ctc_loss_and_grad(logits_ref, om.sparsify_labels(dense, C), lengths)
    g_ref = om.backward(p64, cache, dl_ref, lengths, L)
    eng.zero_grads()
    eng.mini_batch(torch.as_tensor(x).cuda(), torch.as_tensor(lengths).cuda(), torch.as_tensor(dense).cuda())
    assert rel_err(eng.logits.cpu().numpy(), logits_ref) < 2e-4
    np.testing.assert_allclose(eng.loss.cpu().numpy(), loss_ref, rtol=1e-3, atol=1e-5)
    g = eng.to_numpy(eng.grads)
    for k in g_ref:
        assert rel_err(g[k], g_ref[k]) < 5e-3, k
    # the exact-f32 engine on the same inputs must agree with it to the split-precision level
    ref = Engine(L, H, D, C, B, T, U, seed=7)
    ref.forward(torch.as_tensor(x).cuda(), torch.as_tensor(lengths).cuda())
    assert rel_err(eng.logits.cpu().numpy(), ref.logits.cpu().numpy()) < 2e-4


@pytest.mark.parametrize("L,H,D,C,B,T,U", [(3, 512, 40, 80, 32, 16, 8), (2, 256, 40, 80, 20, 40, 12), (1, 128, 40, 80, 2, 101, 40),
                                            (2, 1024, 120, 80, 40, 12, 6)],
                         ids=["dataflow-512", "dataflow-256", "step-kernels-128-run-bf16x3", "per-layer-1024"])
def test_bf16_option_parity(L, H, D, C, B, T, U):
    """precision='bf16' (opt-in, round 4): every operand of the stack's products rounded to ONE bf16, one MFMA per product, f32
    accumulation; short sequences here (the error over 998 frames: tests/test_gpu_fullsize_cfg3.py).  Against the float64 oracle
    with bf16-sized bounds, and it must actually BE a reduced-precision path where the dataflow / per-layer kernels run it."""
    from rnn_speech_amd.engine import Engine
    eng = Engine(L, H, D, C, B, T, U, seed=7, precision="bf16")
    x, lengths, dense = make_batch(T, B, D, C, U, seed=L * 100 + H)
    p64 = {k: v.astype(np.float64) for k, v in eng.to_numpy().items()}
    logits_ref, final_ref, cache = om.forward(p64, x.astype(np.float64), lengths, L, keep_cache=True)
    loss_ref, dl_ref = om.ctc_loss_and_grad(logits_ref, om.sparsify_labels(dense, C), lengths)
    g_ref = om.backward(p64, cache, dl_ref, lengths, L)
    eng.zero_grads()
    eng.mini_batch(torch.as_tensor(x).cuda(), torch.as_tensor(lengths).cuda(), torch.as_tensor(dense).cuda())
    e = rel_err(eng.logits.cpu().numpy(), logits_ref)
    assert e < 1e-2, e
    if H != 128:
        assert e > 2e-5, e          # (H = 128 with a reduced precision is outside the dataflow kernels: bf16x3 step kernels)
    np.testing.assert_allclose(eng.loss.cpu().numpy(), loss_ref, rtol=5e-3, atol=1e-4)
    g = eng.to_numpy(eng.grads)
    for k in g_ref:
        assert rel_err(g[k], g_ref[k]) < 3e-2, (k, rel_err(g[k], g_ref[k]))
    eng.check()


@pytest.mark.parametrize("L,H,B,T", [(2, 128, 20, 30), (2, 64, 5, 21), (1, 256, 33, 70), (2, 1024, 20, 12)],
                         ids=["dataflow", "step-kernels", "dataflow-workers", "per-layer-1024"])
def test_bidirectional_forward_backward_parity(L, H, B, T):
    """Bidirectional option (BASELINE configs[4]; no reference counterpart): a second stack over tf.reverse_sequence'd input,
    top outputs concatenated.  Logits, CTC loss and EVERY gradient tensor against the float64 oracle; ragged lengths with a
    zero-length row; a second mini-batch accumulates."""
    from rnn_speech_amd.engine import Engine
    D, C, U = 40, 80, 8
    eng = Engine(L, H, D, C, B, T, U, seed=9, bidirectional=True)
    rng = np.random.RandomState(4)
    p = eng.to_numpy()
    assert "bw_kernel_%d" % (L - 1) in p and p["output_w"].shape == (2 * H, C)
    for k in p:
        if p[k].ndim == 1:
            p[k] = (rng.randn(*p[k].shape) * 0.1).astype(np.float32)
    eng.load_numpy(p)
    x, lengths, dense = make_batch(T, B, D, C, U, seed=H + B)
    p64 = {k: v.astype(np.float64) for k, v in p.items()}
    logits_ref, cache = om.forward_bidirectional(p64, x.astype(np.float64), lengths, L)
    loss_ref, dl_ref = om.ctc_loss_and_grad(logits_ref, om.sparsify_labels(dense, C), lengths)
    g_ref = om.backward_bidirectional(p64, cache, dl_ref, lengths, L)
    dx, dlen, dlab = torch.as_tensor(x).cuda(), torch.as_tensor(lengths).cuda(), torch.as_tensor(dense).cuda()
    with eng.on_stream():
        eng.zero_grads()
        eng.mini_batch(dx, dlen, dlab)
    torch.cuda.synchronize()
    eng.check()
    assert rel_err(eng.logits.cpu().numpy(), logits_ref) < 1e-4
    np.testing.assert_allclose(eng.loss.cpu().numpy(), loss_ref, rtol=1e-3, atol=1e-5)
    g = eng.to_numpy(eng.grads)
    for k in g_ref:
        assert rel_err(g[k], g_ref[k]) < 2e-3, k
    with eng.on_stream():
        eng.mini_batch(dx, dlen, dlab)                     # accumulates
    torch.cuda.synchronize()
    g2 = eng.to_numpy(eng.grads)
    for k in g_ref:
        assert rel_err(g2[k], 2.0 * g_ref[k]) < 2e-3, k


@pytest.mark.parametrize("H", [128, 512])
def test_training_cycle_arms_the_hand_off_panels(H):
    """Engine.mini_batch runs lstm_fwd with AMDSPEECH_LSTM_ARM_NEXT: the backward call's panels are filled beside the forward
    kernel and the forward panels again behind it, and the following calls of the same layout skip their fills (ARMED).
    Six optimiser steps that mix lengths (another layout on the same allocation in between), a forward without a backward and
    an inference forward give the same losses and parameters as the same steps with every call filling its own panels.
    H = 512 (round 5): the forward kernel's x-product workers hand their tiles over through a history tagged with the LAUNCH's
    parity -- flipped from call to call (ARMED / AMDSPEECH_LSTM_SAME_WS), re-tagged only where a longer sequence follows a
    shorter one (48 -> 31 -> 48 -> 40 frames here), zeroed by every call when the flags are off."""
    from rnn_speech_amd import ops
    from rnn_speech_amd.engine import Engine
    L, D, C, B, T, U = 3, 20, 80, 20, 48, 10
    batches = [make_batch(T, B, D, C, U, seed=30 + i, full=(i % 2 == 0)) for i in range(6)]

    def run(arm):
        old, ops._ARM = ops._ARM, arm
        try:
            eng = Engine(L, H, D, C, B, T, U, seed=13)
            losses, flags = [], []
            for i, (x, lengths, dense) in enumerate(batches):
                if i == 3:        # a ragged batch: the prefix layout of its longest utterance, on the same allocation
                    lengths = np.minimum(lengths, 31).astype(np.int32)
                if i == 5:
                    lengths = np.minimum(lengths, 40).astype(np.int32)
                dx, dl, dd = torch.as_tensor(x).cuda(), torch.as_tensor(lengths).cuda(), torch.as_tensor(dense).cuda()
                if i == 2:        # a forward that no backward follows, and an inference forward
                    eng.mini_batch(dx, dl, dd, compute_gradients=False)
                    eng.forward(dx, dl)
                eng.zero_grads()
                eng.mini_batch(dx, dl, dd, 0.9, 0.8, seed=i + 1, max_len=int(lengths.max()))
                flags.append(eng.lstm_ws._armed is not None)
                eng.apply(1e-3, 1.0)
                losses.append(eng.loss.cpu().numpy().copy())
            eng.check()
            return np.stack(losses), eng.params.cpu().numpy().copy(), flags
        finally:
            ops._ARM = old

    la, pa, fa = run(True)
    lb, pb, fb = run(False)
    assert all(fa) and not any(fb)          # (the state machine of ops.lstm_fwd / lstm_bwd was exercised, and switched off)
    assert np.all(np.isfinite(la)) and np.all(la[:, 0] > 0)
    # (H = 512, the losses too: run-to-run parameter noise of 1e-5 after a few Adam steps comes back as 1e-5 of a later loss)
    np.testing.assert_allclose(la, lb, rtol=1e-5 if H == 128 else 1e-4)
    # (H = 512: the split-K weight gradients of 2L x 128 tiles meet in f32 atomics -- two runs of the SAME variant differ by 1e-5 of
    #  the largest parameter after six Adam steps; the tags themselves are 1 ulp of a partial pre-activation)
    assert np.abs(pa - pb).max() < (1e-5 if H == 128 else 1e-4) * np.abs(pb).max()


def test_side_work_beside_the_forward_recurrence():
    """Engine.mini_batch(beside_forward=hook): with the whole-sequence forward kernel and XCDs to spare the hook is called with the
    ordering point IN FRONT of that launch (ops.lstm_beside_forward > 0) and its work -- here the front end of another batch, whose
    frame kernel is a work queue -- runs while the recurrence does; without such a launch (H = 1024: per-layer kernels, every XCD
    busy) the same hook is called in the slot beside the CTC stage.  Either way: one call per mini-batch, the step's own numbers
    unchanged, the side work's results intact, and the backward kernel ordered behind it."""
    from rnn_speech_amd import engine as eng_mod
    from rnn_speech_amd import ops
    from rnn_speech_amd.engine import Engine
    rng = np.random.RandomState(3)
    pcm = torch.as_tensor((rng.randn(4, 16000) * 0.1).astype(np.float32)).cuda()
    n_samples = [16000, 12000, 16000, 9000]
    ref_feat = ops.frontend(pcm, n_samples, 16000, "mfcc", 101, 40)[0].cpu().numpy()
    side = torch.cuda.Stream()
    # (H = 512, round 5: the spare XCDs run the forward kernel's x-product workers.  With the separate CTC launches nothing is
    #  reported idle -- the side work goes beside the CTC stage; with the CTC head inside the LSTM launches there is no such stage,
    #  and the workgroups the forward kernel keeps in reserve on the spare XCDs take it again: 2)
    for (L, H, D, C, B, T, U), idle in (((3, 128, 40, 80, 20, 40, 10), 2), ((2, 1024, 40, 80, 16, 8, 4), 0),
                                        ((3, 512, 40, 80, 20, 12, 4), 2 if eng_mod._FUSED_CTC else 0)):
        x, lengths, dense = make_batch(T, B, D, C, U, seed=77, full=True)
        dx, dl, dd = torch.as_tensor(x).cuda(), torch.as_tensor(lengths).cuda(), torch.as_tensor(dense).cuda()
        results = {}
        for mode in ("plain", "hooked", "hooked-late"):
            eng = Engine(L, H, D, C, B, T, U, seed=5)
            calls = []

            def hook(after):
                calls.append(after)
                side.wait_event(after)
                with torch.cuda.stream(side):
                    f = ops.frontend(pcm, n_samples, 16000, "mfcc", 101, 40)[0]
                    ev = torch.cuda.Event()
                    ev.record(side)
                calls.append(f)
                return ev

            old = eng_mod._BESIDE_FORWARD
            eng_mod._BESIDE_FORWARD = mode != "hooked-late"
            try:
                with eng.on_stream():
                    eng.zero_grads()
                    eng.mini_batch(dx, dl, dd, 0.9, 0.8, seed=4, beside_forward=None if mode == "plain" else hook)
                    placed = ops.lstm_beside_forward(eng._ws, side)
                torch.cuda.synchronize()
            finally:
                eng_mod._BESIDE_FORWARD = old
            eng.check()
            assert placed == idle, (mode, placed)             # (3 layers x 2 batch tiles = 6 of 8 XCDs; H = 1024: none to spare)
            if mode != "plain":
                assert len(calls) == 2                        # one call per mini-batch
                assert np.abs(calls[1].cpu().numpy() - ref_feat).max() < 1e-5
            results[mode] = (eng.loss.cpu().numpy().copy(), eng.to_numpy(eng.grads))
        for mode in ("hooked", "hooked-late"):
            np.testing.assert_allclose(results[mode][0], results["plain"][0], rtol=1e-5)
            for k, g in results["plain"][1].items():
                # (split-K weight gradients accumulate with atomics: run-to-run differences in the last bits)
                assert np.abs(results[mode][1][k] - g).max() <= 2e-5 * (np.abs(g).max() + 1e-30), (mode, k)


@pytest.mark.parametrize("where", ["forward", "backward"])
@pytest.mark.parametrize("L,H,B,T", [(3, 128, 20, 40), (3, 512, 32, 24), (2, 1024, 20, 12)], ids=["H128", "H512-x-workers", "H1024-per-layer"])
def test_dataflow_time_out_is_survived(L, H, B, T, where):
    """VERDICT r4 #6 / r5 #2c.  AMDSPEECH_LSTM_INJECT_TIMEOUT makes ONE whole-sequence launch -- the forward one, or the BACKWARD one
    (lstm_bwd_flow2 with the CTC leader, the weight-gradient workers and dZ_0 inside: the launch with the most waits; lstm_bwd_big
    at 1024 units) -- give up on its first unsatisfied wait
    (what a launch whose workgroups are not all resident does after its limit): amdspeech_lstm_status reports it, the mini-batch's
    results are garbage.  The way out the drop-in class takes (acoustic_model.run_step): take the gradient contribution back, run
    the mini-batch again with AMDSPEECH_LSTM_PER_DIAGONAL -- logits, loss and EVERY gradient tensor match the float64 oracle, an
    accumulated earlier mini-batch is still there -- and the NEXT call is back on the dataflow kernels (hand-off panels and, at
    H = 512, the x-product workers' tile history re-initialised), again in agreement with the oracle."""
    from rnn_speech_amd.engine import Engine
    D, C, U = 20, 80, 8
    eng = Engine(L, H, D, C, B, T, U, seed=7)
    p = eng.to_numpy()
    p64 = {k: v.astype(np.float64) for k, v in p.items()}

    def oracle(seed):
        x, lengths, dense = make_batch(T, B, D, C, U, seed=seed)
        lg, _, cache = om.forward(p64, x.astype(np.float64), lengths, L, keep_cache=True)
        loss, dl = om.ctc_loss_and_grad(lg, om.sparsify_labels(dense, C), lengths)
        return (torch.as_tensor(x).cuda(), torch.as_tensor(lengths).cuda(), torch.as_tensor(dense).cuda()), lg, loss, \
            om.backward(p64, cache, dl, lengths, L)

    b1, lg1, loss1, g1 = oracle(41)
    b2, lg2, loss2, g2 = oracle(42)
    b3, lg3, loss3, g3 = oracle(43)
    with eng.on_stream():
        eng.zero_grads()
        eng.mini_batch(*b1)                                        # a healthy mini-batch accumulates first
        eng.check()                                                # (raises with the launch's error flags)
        kept = eng.grads.clone()
        if where == "forward":
            eng.lstm_ws._inject_timeout = 1
        else:
            eng.lstm_ws._inject_timeout_bwd = 1
        eng.mini_batch(*b2)
        torch.cuda.synchronize()
        assert not eng.healthy()                                   # reported, not hung (and not fatal)
        from rnn_speech_amd import lib as _lib
        with pytest.raises(_lib.DataflowTimeout):                  # ... as a time-out (the recoverable kind), with the launch named
            eng.check()
        eng.grads.copy_(kept)                                      # its contribution is taken back ...
        eng.mini_batch(*b2, per_diagonal=True)                     # ... and the mini-batch repeated on the launch-per-diagonal kernels
        torch.cuda.synchronize()
        assert eng.healthy()
        assert rel_err(eng.logits.cpu().numpy(), lg2) < 1e-4
        np.testing.assert_allclose(eng.loss.cpu().numpy(), loss2, rtol=1e-3, atol=1e-5)
        g = eng.to_numpy(eng.grads)
        for k in g1:
            assert rel_err(g[k], g1[k] + g2[k]) < 2e-3, k
        eng.zero_grads()
        eng.mini_batch(*b3)                                        # training continues on the dataflow kernels
        torch.cuda.synchronize()
        assert eng.healthy()
        assert rel_err(eng.logits.cpu().numpy(), lg3) < 1e-4
        np.testing.assert_allclose(eng.loss.cpu().numpy(), loss3, rtol=1e-3, atol=1e-5)
        g = eng.to_numpy(eng.grads)
        for k in g3:
            assert rel_err(g[k], g3[k]) < 2e-3, k


def test_reverse_sequences_matches_oracle():
    from rnn_speech_amd import ops
    rng = np.random.RandomState(0)
    T, B, H = 37, 9, 24
    x = rng.randn(T, B, H).astype(np.float32)
    lens = np.array([37, 0, 5, 1, 36, 20, 50, 2, 19], np.int32)       # incl. 0 and an untruncated length > T
    got = ops.reverse_sequences(torch.as_tensor(x).cuda(), torch.as_tensor(lens).cuda()).cpu().numpy()
    assert np.array_equal(got, om.reverse_sequences(x, lens))
    acc = torch.ones(T, B, H, device="cuda")
    ops.reverse_sequences(torch.as_tensor(x).cuda(), torch.as_tensor(lens).cuda(), out=acc, accumulate=True)
    assert np.allclose(acc.cpu().numpy(), om.reverse_sequences(x, lens) + 1.0)


@pytest.mark.parametrize("env", [{"AMDSPEECH_FLOW_DZ0": "0"}, {"AMDSPEECH_FLOW": "0"}, {"AMDSPEECH_BIG": "0"},
                                 {"AMDSPEECH_GEMM_DIRECT": "0", "AMDSPEECH_GEMM_KC_DIRECT": "0"}, {"AMDSPEECH_FLOW_FWD_WORKERS": "0"},
                                 {"AMDSPEECH_FLOW_CTC": "0"}, {"AMDSPEECH_FLOW_WORKER_DEAL": "1"}, {"AMDSPEECH_FLOW_WORKER_DEAL": "0"}],
                         ids=["dz0-gemm-after-the-kernel", "launch-per-diagonal", "no-per-layer-1024", "lds-gemm-only",
                              "forward-without-x-workers", "ctc-stage-as-separate-launches", "worker-tiles-always-dealt",
                              "worker-tiles-never-dealt"])
def test_non_default_kernel_choices_keep_parity(env):
    """The switches of INTEGRATION.md select kernels that the default path no longer runs (the library reads them once per
    process): the dataflow-shaped parity cases again, in a child process per switch."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_gpu_model.py"), "-m", "gpu", "-q", "-x", "-k",
                          "test_forward_backward_adam_parity or dataflow-kernels or test_training_cycle_arms"],
                         env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
