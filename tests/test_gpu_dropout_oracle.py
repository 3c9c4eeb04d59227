"""The DROPOUT-ON configuration (config.ini: input_keep_prob 0.8, output_keep_prob 0.5 -- what the headline number is
benchmarked on) against the float64 oracle.

Reference semantics (models/AcousticModel.py:227-233, :648-652): every layer's cell sits in a
tf.contrib.rnn.DropoutWrapper(cell, input_keep_prob, output_keep_prob): the layer's INPUT and its OUTPUT are multiplied by
independent inverted-dropout masks (Bernoulli(keep) / keep), fresh per layer and side; the recurrent state is not masked.

The kernels draw their masks from a counter-based generator (csrc/common.h uniform01, csrc/lstm.hip zmult), so the random
stream cannot equal TensorFlow's -- what CAN be checked, and is here, is that (1) the masks are what the generator's published
definition says (restated in numpy below), Bernoulli(keep)/keep and independent between sides and layers, and (2) with THOSE
masks fed to the oracle's DropoutWrapper restatement (oracle.model.forward/backward in_masks/out_masks) the logits, CTC losses
and every gradient tensor agree -- on every kernel family, in both precisions, bidirectional, and at the full headline size with
the in-kernel GEMM workers active.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import model as om  # noqa: E402  (checker only)


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


# ---- numpy restatement of the generator (csrc/common.h: mix32 / uniform01; csrc/lstm.hip: zmult)
def _mix32(x):
    x = x.astype(np.uint32)
    x ^= x >> np.uint32(16)
    x = (x * np.uint32(0x7feb352d)).astype(np.uint32)
    x ^= x >> np.uint32(15)
    x = (x * np.uint32(0x846ca68b)).astype(np.uint32)
    x ^= x >> np.uint32(16)
    return x


def restated_multipliers(seed, stream, n, keep):
    """mask/keep of elements 0..n-1 of dropout stream `stream` (2*layer for the input side, 2*layer + 1 for the output side)."""
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint32)
        a = _mix32(idx ^ np.uint32(seed & 0xffffffff))
        b = _mix32((a + np.uint32((stream * 0x9e3779b9) & 0xffffffff) + np.uint32((seed >> 32) & 0xffffffff)).astype(np.uint32))
    u = (b >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    return np.where(u < np.float32(keep), np.float32(1.0) / np.float32(keep), np.float32(0.0)).astype(np.float32)


def engine_masks(ws, L):
    """(in_masks, out_masks) as float64 numpy [T,B,H] per layer, exported by the library for the descriptor of `ws`."""
    from rnn_speech_amd import ops
    ins = [ops.lstm_dropout_multipliers(ws, "in", l).cpu().numpy().astype(np.float64) for l in range(L)]
    outs = [ops.lstm_dropout_multipliers(ws, "out", l).cpu().numpy().astype(np.float64) for l in range(L)]
    return ins, outs


def make_batch(T, B, D, C, U, seed, full=False):
    rng = np.random.RandomState(seed)
    x = rng.randn(T, B, D).astype(np.float32)
    lengths = np.full(B, T, np.int32) if full else rng.randint(max(2, T // 2), T + 1, size=B).astype(np.int32)
    if B > 2 and not full:
        lengths[1] = 0
    dense = np.zeros((B, U), np.int32)
    for b in range(B):
        n = rng.randint(1, max(2, min(U - 1, int(lengths[b]) // 3 + 1)))
        dense[b, :n] = rng.randint(1, C - 1, size=n)
        dense[b, n] = C - 1
    return x, lengths, dense


def test_exported_multipliers_are_the_documented_generator_and_independent():
    from rnn_speech_amd import ops
    T, B, H, L = 40, 7, 128, 3
    seed = (0x1234abcd << 32) | 0x9e3779b1
    ws = ops.LstmWorkspace(T, B, H, L, keep_in=0.8, keep_out=0.5, seed=seed)
    n = T * B * H
    ins, outs = engine_masks(ws, L)
    for l in range(L):
        assert np.array_equal(ins[l].ravel().astype(np.float32), restated_multipliers(seed, 2 * l, n, 0.8)), l
        assert np.array_equal(outs[l].ravel().astype(np.float32), restated_multipliers(seed, 2 * l + 1, n, 0.5)), l
        assert set(np.unique(ins[l])) == {0.0, float(np.float32(1.0) / np.float32(0.8))}
        assert set(np.unique(outs[l])) == {0.0, 2.0}
        assert abs((ins[l] != 0).mean() - 0.8) < 0.01 and abs((outs[l] != 0).mean() - 0.5) < 0.01
    # independence: every pair of the 2L masks (in particular the output mask of layer l and the input mask of layer l+1, which
    # multiply the SAME tensor Z_{l+1}) is uncorrelated -- |corr| of n = 35,840 Bernoulli pairs stays within 5 sigma = 0.027
    flat = [(m != 0).ravel().astype(np.float64) for m in ins + outs]
    for i in range(len(flat)):
        for j in range(i + 1, len(flat)):
            assert abs(np.corrcoef(flat[i], flat[j])[0, 1]) < 0.027, (i, j)
    # another seed: other masks; keep = 1: all ones
    ws.set_dropout(0.8, 0.5, seed + 1)
    assert not np.array_equal(ops.lstm_dropout_multipliers(ws, "in", 0).cpu().numpy(), ins[0])
    ws.set_dropout(1.0, 1.0, seed)
    assert float(ops.lstm_dropout_multipliers(ws, "out", 1).min()) == 1.0


# L, H, D, C, B, T, U, precision -- every kernel family
SHAPES = [
    (2, 64, 20, 80, 5, 25, 10, "f32"),         # launch-per-diagonal kernels
    (3, 48, 40, 80, 33, 30, 12, "f32"),        # launch-per-diagonal, H = 48, three batch blocks
    (3, 128, 40, 80, 20, 40, 12, "f32"),       # dataflow kernels, ragged second batch tile
    (2, 256, 40, 80, 64, 70, 16, "f32"),       # dataflow kernels, all 8 XCDs carry a group, T >= 64: in-kernel GEMM workers
    (3, 512, 40, 80, 32, 24, 8, "f32"),        # BASELINE configs[1] shape, short in time
    (2, 1024, 120, 80, 40, 14, 6, "f32"),      # per-layer H = 1024 kernels (BASELINE configs[2] family)
    (3, 128, 40, 80, 20, 40, 12, "bf16x3"),    # split precision inside the dataflow kernels
    (2, 1024, 120, 80, 40, 14, 6, "bf16x3"),   # split precision inside the per-layer kernels
    (2, 64, 20, 80, 5, 25, 10, "bf16x3"),      # split precision, launch-per-diagonal kernels
]


@pytest.mark.parametrize("L,H,D,C,B,T,U,precision", SHAPES)
def test_dropout_on_training_step_matches_oracle(L, H, D, C, B, T, U, precision):
    from rnn_speech_amd.engine import Engine
    eng = Engine(L, H, D, C, B, T, U, seed=7, precision=precision)
    rng = np.random.RandomState(2)
    p = eng.to_numpy()
    for k in p:
        if p[k].ndim == 1:
            p[k] = (rng.randn(*p[k].shape) * 0.1).astype(np.float32)
    eng.load_numpy(p)
    x, lengths, dense = make_batch(T, B, D, C, U, seed=L * 100 + H)
    dx, dlen, dlab = torch.as_tensor(x).cuda(), torch.as_tensor(lengths).cuda(), torch.as_tensor(dense).cuda()
    keep_in, keep_out, seed = 0.8, 0.5, 4242
    with eng.on_stream():
        eng.zero_grads()
        eng.mini_batch(dx, dlen, dlab, keep_in, keep_out, seed=seed)
    torch.cuda.synchronize()
    eng.check()
    in_masks, out_masks = engine_masks(eng._ws, L)          # (the descriptor still holds this step's keep / seed)

    p64 = {k: v.astype(np.float64) for k, v in p.items()}
    logits_ref, final_ref, cache = om.forward(p64, x.astype(np.float64), lengths, L, keep_cache=True,
                                              in_masks=in_masks, out_masks=out_masks)
    loss_ref, dl_ref = om.ctc_loss_and_grad(logits_ref, om.sparsify_labels(dense, C), lengths)
    g_ref = om.backward(p64, cache, dl_ref, lengths, L, in_masks=in_masks, out_masks=out_masks)
    tol_logits, tol_grad = (1e-4, 2e-3) if precision == "f32" else (2e-4, 5e-3)
    assert rel_err(eng.logits.cpu().numpy(), logits_ref) < tol_logits
    np.testing.assert_allclose(eng.loss.cpu().numpy(), loss_ref, rtol=1e-3, atol=1e-5)
    # the state is NOT masked: final (c, h) of every layer equal the oracle's
    h, c = eng.final_state()
    for l in range(L):
        assert np.abs(c[l].cpu().numpy() - final_ref[l][0]).max() < 2e-4
        assert np.abs(h[l].cpu().numpy() - final_ref[l][1]).max() < 2e-4
    g = eng.to_numpy(eng.grads)
    for k in g_ref:
        assert rel_err(g[k], g_ref[k]) < tol_grad, (k, rel_err(g[k], g_ref[k]))
    # and the masks matter: the same step without them is a different function
    no_mask, _, _ = om.forward(p64, x.astype(np.float64), lengths, L)
    assert rel_err(no_mask, logits_ref) > 1e-2


@pytest.mark.parametrize("L,H,B,T", [(2, 128, 20, 30), (2, 64, 5, 21), (2, 1024, 20, 12)],
                         ids=["dataflow", "step-kernels", "per-layer-1024"])
def test_dropout_on_bidirectional_matches_oracle(L, H, B, T):
    """The two stacks of the bidirectional option have their own dropout streams (the backward-direction stack's masks are
    indexed in ITS time, i.e. on the reversed sequence)."""
    from rnn_speech_amd.engine import Engine
    D, C, U = 40, 80, 8
    eng = Engine(L, H, D, C, B, T, U, seed=9, bidirectional=True)
    x, lengths, dense = make_batch(T, B, D, C, U, seed=H + B)
    p64 = {k: v.astype(np.float64) for k, v in eng.to_numpy().items()}
    dx, dlen, dlab = torch.as_tensor(x).cuda(), torch.as_tensor(lengths).cuda(), torch.as_tensor(dense).cuda()
    with eng.on_stream():
        eng.zero_grads()
        eng.mini_batch(dx, dlen, dlab, 0.8, 0.5, seed=77)
    torch.cuda.synchronize()
    eng.check()
    mf, mb = engine_masks(eng._ws, L), engine_masks(eng._ws_b, L)
    assert not np.array_equal(mf[0][0], mb[0][0])           # two streams
    logits_ref, cache = om.forward_bidirectional(p64, x.astype(np.float64), lengths, L, masks_fw=mf, masks_bw=mb)
    loss_ref, dl_ref = om.ctc_loss_and_grad(logits_ref, om.sparsify_labels(dense, C), lengths)
    g_ref = om.backward_bidirectional(p64, cache, dl_ref, lengths, L)
    assert rel_err(eng.logits.cpu().numpy(), logits_ref) < 1e-4
    np.testing.assert_allclose(eng.loss.cpu().numpy(), loss_ref, rtol=1e-3, atol=1e-5)
    g = eng.to_numpy(eng.grads)
    for k in g_ref:
        assert rel_err(g[k], g_ref[k]) < 2e-3, (k, rel_err(g[k], g_ref[k]))


def test_dropout_on_headline_configuration_matches_oracle_at_full_size():
    """BASELINE configs[1] exactly as bench.py runs it -- 3x512, D = 40, B = 32, T = 1001, keep 0.8 / 0.5, on the engine's own
    stream (in-kernel GEMM workers and the host-launched remainder both active) -- against the float64 oracle with the kernels'
    own masks.  Two utterances are live (the others have length 0): utterances are independent and the gradient is a sum over
    them, so the flat gradient must equal the oracle's for the 2-utterance sub-batch.  Twice, so every ring slot is reused."""
    from rnn_speech_amd.engine import Engine
    L, H, D, C, B, T, U = 3, 512, 40, 80, 32, 1001, 161
    eng = Engine(L, H, D, C, B, T, U, seed=1234)
    rng = np.random.RandomState(3)
    p = eng.to_numpy()
    for k in p:
        if p[k].ndim == 1:
            p[k] = (rng.randn(*p[k].shape) * 0.1).astype(np.float32)
    eng.load_numpy(p)
    x = rng.randn(T, B, D).astype(np.float32)
    sel = [3, 21]                                           # one utterance in each 16-row batch tile
    live = np.zeros(B, np.int32)
    live[3], live[21] = T, 733
    dense = np.zeros((B, U), np.int32)
    for b in range(B):
        n = rng.randint(80, 161)
        dense[b, :n - 1] = rng.randint(1, C - 1, size=n - 1)
        dense[b, n - 1] = C - 1
    dx, dlen, dlab = torch.as_tensor(x).cuda(), torch.as_tensor(live).cuda(), torch.as_tensor(dense).cuda()
    p64 = {k: v.astype(np.float64) for k, v in p.items()}
    for rep in range(2):
        seed = 1000 + rep
        with eng.on_stream():
            eng.zero_grads()
            eng.mini_batch(dx, dlen, dlab, 0.8, 0.5, seed=seed)
        torch.cuda.synchronize()
        eng.check()
        assert eng._head is not None             # the headline path: CTC stage inside the two whole-sequence launches
        ins, outs = engine_masks(eng._ws, L)
        ins = [m[:, sel, :] for m in ins]
        outs = [m[:, sel, :] for m in outs]
        logits_ref, _, cache = om.forward(p64, x[:, sel, :].astype(np.float64), live[sel], L, keep_cache=True,
                                          in_masks=ins, out_masks=outs)
        loss_ref, dl_ref = om.ctc_loss_and_grad(logits_ref, om.sparsify_labels(dense[sel], C), live[sel])
        g_ref = om.backward(p64, cache, dl_ref, live[sel], L, in_masks=ins, out_masks=outs)
        assert rel_err(eng.logits.cpu().numpy()[:, sel, :], logits_ref) < 1e-4
        np.testing.assert_allclose(eng.loss.cpu().numpy()[sel], loss_ref, rtol=1e-3)
        g = eng.to_numpy(eng.grads)
        # (a) BPTT alone: the oracle's backward pass fed with the GPU's OWN dlogits -- isolates the dropout-on recurrence,
        # the weight-gradient products and the masks from the CTC arithmetic; f32 MFMA accumulation over 1001 frames
        dl_gpu = eng.dlogits.cpu().numpy()[:, sel, :].astype(np.float64)
        g_bptt = om.backward(p64, cache, dl_gpu, live[sel], L, in_masks=ins, out_masks=outs)
        for k in g_bptt:
            assert rel_err(g[k], g_bptt[k]) < 2e-4, (rep, k, rel_err(g[k], g_bptt[k]))
        # (b) end to end.  The logits agree to 1e-6; what is left is the CTC gradient itself.  Rounds 1-3 kept alpha / beta as f32
        # log-space sums like TensorFlow's op (|alpha| ~ 1e3 after 1001 frames, ulp 6e-5, a random walk over T frames), which put
        # dlogits 2-3e-3 of its maximum away from the float64 oracle (bounds 5e-3 / 4e-3 here in round 3).  Round 4: the recursion
        # state and the gradient's exponent are float64 (csrc/ctc.hip) -- the north_star-level bounds hold again
        print("dlogits rel err %.2e" % rel_err(dl_gpu, dl_ref))
        assert rel_err(dl_gpu, dl_ref) < 1e-3
        for k in g_ref:
            assert rel_err(g[k], g_ref[k]) < 2e-3, (rep, k, rel_err(g[k], g_ref[k]))
