"""The CTC head INSIDE the whole-sequence LSTM launches (amdspeech_lstm_fwd_ctc / amdspeech_lstm_bwd_ctc, csrc/ctc_flow.h) against the
same mini-batch run with the CTC stage as separate launches (output Linear, log-softmax, alpha / beta, gradient, dlogits . W_o^T), and
against the staged CTC call on the fused path's own logits (the loss must be BIT-identical: same device code, same logits).
Reference semantics: /root/reference/models/AcousticModel.py:241-247 (output layer), :356-357 (tf.nn.ctc_loss,
ignore_longer_outputs_than_inputs) -- the oracle parity of the whole step is test_gpu_model.py / test_gpu_fullsize.py, which run the
fused path wherever the shape takes it."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def make_batch(T, B, D, C, U, seed, full=False):
    rng = np.random.RandomState(seed)
    x = rng.randn(T, B, D).astype(np.float32)
    lengths = np.full(B, T, np.int32) if full else rng.randint(min(max(2, T // 2), T), T + 1, size=B).astype(np.int32)
    dense = np.zeros((B, U), np.int32)
    for b in range(B):
        n = rng.randint(1, max(2, min(U - 1, int(lengths[b]) // 3 + 1)))
        dense[b, :n] = rng.randint(1, C - 1, size=n)
        dense[b, n] = C - 1
    if B > 3 and not full:
        lengths[1] = 0                               # a padded row of a short final batch: loss 0, no gradient
        dense[2, :] = 0                              # an empty label row (the reference feeds [C-1] for it)
        dense[3, :U - 1] = 5                         # more labels than frames allow: invalid, loss 0 (ignore_longer_outputs_than_inputs)
        lengths[3] = max(1, min(T, (U - 1) // 2))
    return x, lengths, dense


def run(fused, L, H, D, C, B, T, U, seed, keep=(1.0, 1.0), full=False, max_len=None):
    from rnn_speech_amd import engine as E
    old, E._FUSED_CTC = E._FUSED_CTC, fused
    try:
        eng = E.Engine(L, H, D, C, B, T, U, seed=21)
        x, lengths, dense = make_batch(T, B, D, C, U, seed, full=full)
        if max_len is not None:
            lengths = np.minimum(lengths, max_len).astype(np.int32)
        dx, dl, dd = torch.as_tensor(x).cuda(), torch.as_tensor(lengths).cuda(), torch.as_tensor(dense).cuda()
        eng.zero_grads()
        eng.mini_batch(dx, dl, dd, keep[0], keep[1], seed=5, max_len=max_len)
        torch.cuda.synchronize()
        eng.check()
        took = eng._head is not None
        out = dict(logits=eng.logits.cpu().numpy().copy(), loss=eng.loss.cpu().numpy().copy(), dlogits=eng.dlogits.cpu().numpy().copy(),
                   grads=eng.grads.cpu().numpy().copy(), dz0=eng._ws.dz0.cpu().numpy().copy(), took=took, lengths=lengths)
        # the staged CTC call on THESE logits, into fresh buffers
        from rnn_speech_amd import ops
        Tr = eng._Tr
        loss2 = torch.zeros_like(eng.loss)
        dlog2 = torch.empty_like(eng.dlogits[:Tr])
        ops.ctc_loss_fwd_bwd(eng.logits[:Tr].clone(), dd, dl, ws=ops.CtcWorkspace(Tr, B, C, U), loss=loss2, dlogits=dlog2)
        torch.cuda.synchronize()
        out["staged_loss"] = loss2.cpu().numpy().copy()
        out["staged_dlogits"] = dlog2.cpu().numpy().copy()
        out["Tr"] = Tr
        return out
    finally:
        E._FUSED_CTC = old


SHAPES = [
    # L, H, D, C, B, T, U
    (2, 128, 20, 80, 3, 20, 8),        # the smoke shape: six spare XCDs, one utterance per team
    (3, 512, 40, 80, 32, 50, 12),      # the headline shape, short: x-product workers beside the followers, two utterances per team, T % 16 != 0
    (1, 128, 20, 80, 100, 70, 20),     # seven batch tiles of one layer: 100 utterances on 8 follower workgroups' 16 teams -> does NOT fuse
    (2, 256, 40, 80, 20, 33, 16),      # ragged second batch tile, 33 frames = two chunks + one frame
    (3, 128, 20, 80, 20, 1, 2),        # a single frame
    (3, 512, 40, 80, 17, 130, 60),     # 121 extended states per utterance (two waves of the chain busy), nine chunks
    (3, 512, 40, 80, 32, 240, 70),     # 141 extended states: the staged call runs ctc_alpha_beta3_kernel -- the recursion the head restates
]


def same_recursion(U):
    """The staged call picks ctc_alpha_beta3_kernel (float64 state, DPP shift) for 129 .. 384 extended states; the fused head always runs
    that recursion.  Only there are the two losses the same bits; with shorter targets the staged call keeps a float32 state."""
    return 128 < 2 * U + 1 <= 384


def check_loss_against_staged(a, U):
    if same_recursion(U):
        assert np.array_equal(a["loss"], a["staged_loss"])
    else:
        np.testing.assert_allclose(a["loss"], a["staged_loss"], rtol=3e-6, atol=1e-5)


@pytest.mark.parametrize("L,H,D,C,B,T,U", SHAPES)
def test_fused_head_matches_the_separate_launches(L, H, D, C, B, T, U):
    a = run(True, L, H, D, C, B, T, U, seed=3)
    b = run(False, L, H, D, C, B, T, U, seed=3)
    assert not b["took"]
    if B == 100:
        assert not a["took"]           # (more than two utterances per follower team: the separate launches)
        return
    assert a["took"]
    # logits: another summation order of the same K = H products (K split over four waves)
    scale = np.abs(b["logits"]).max()
    assert np.abs(a["logits"] - b["logits"]).max() < 2e-6 * max(scale, 1.0)
    # the loss: bit-identical to the staged call on the fused path's own logits; and equal to the other path's to rounding
    check_loss_against_staged(a, U)
    np.testing.assert_allclose(a["loss"], b["loss"], rtol=3e-6, atol=1e-5)
    # dlogits: the staged call's, up to the order LDS atomics meet in
    Tr = a["Tr"]
    # (short targets: the staged call's float32 recursion state is the less accurate side -- DESIGN.md 4.3, round 4)
    assert np.abs(a["dlogits"][:Tr] - a["staged_dlogits"]).max() < (2e-6 if same_recursion(U) else 2e-4)
    assert np.abs(a["dlogits"] - b["dlogits"]).max() < 2e-4      # (the other path's logits differ in the last bits; T frames of recursion later ...)
    # ... and everything behind it: dZ_0 and every parameter gradient
    tol = 5e-4
    assert np.abs(a["dz0"] - b["dz0"]).max() < tol * max(np.abs(b["dz0"]).max(), 1e-3) + 1e-7
    assert np.abs(a["grads"] - b["grads"]).max() < tol * max(np.abs(b["grads"]).max(), 1e-3)
    # the conventions: padded row and over-long targets give loss 0 and no gradient; frames past an utterance's end none either
    if B > 3:
        assert a["loss"][1] == 0.0 and not a["dlogits"][:, 1].any()
        assert a["loss"][0] > 0.0 or T < 3       # (a single frame cannot carry a label and its terminator: every row is invalid there)
        if U - 1 > a["lengths"][3]:          # (the over-long row really is over-long at this shape)
            assert a["loss"][3] == 0.0 and not a["dlogits"][:, 3].any()
    for bb in range(B):
        assert not a["dlogits"][a["lengths"][bb]:, bb].any()


def test_fused_head_with_dropout_and_a_shorter_run_length():
    """keep 0.8 / 0.5 (the follower reads the top layer's MASKED output) and max_len < T (the prefix layout of the allocation: the
    logits past the run are the output bias, dlogits zero)."""
    L, H, D, C, B, T, U = 3, 512, 40, 80, 32, 64, 10
    a = run(True, L, H, D, C, B, T, U, seed=9, keep=(0.8, 0.5), max_len=41)
    b = run(False, L, H, D, C, B, T, U, seed=9, keep=(0.8, 0.5), max_len=41)
    assert a["took"] and a["Tr"] == 41
    assert np.abs(a["logits"] - b["logits"]).max() < 2e-6 * max(np.abs(b["logits"]).max(), 1.0)
    check_loss_against_staged(a, U)
    assert np.abs(a["dlogits"] - b["dlogits"]).max() < 2e-4
    assert not a["dlogits"][41:].any()
    assert np.abs(a["grads"] - b["grads"]).max() < 5e-4 * max(np.abs(b["grads"]).max(), 1e-3)


def test_fused_head_over_training_steps():
    """Six optimiser steps, alternating lengths on one allocation (the armed hand-off panels, the two panel sets, the sentinel under
    dZ_top re-filled beside the forward kernel): same losses as the separate launches to rounding."""
    from rnn_speech_amd import engine as E
    L, H, D, C, B, T, U = 3, 512, 40, 80, 32, 48, 10
    batches = [make_batch(T, B, D, C, U, seed=40 + i, full=(i % 2 == 0)) for i in range(6)]

    def steps(fused):
        old, E._FUSED_CTC = E._FUSED_CTC, fused
        try:
            eng = E.Engine(L, H, D, C, B, T, U, seed=13)
            losses = []
            for i, (x, lengths, dense) in enumerate(batches):
                if i == 3:
                    lengths = np.minimum(lengths, 31).astype(np.int32)
                dx, dl, dd = torch.as_tensor(x).cuda(), torch.as_tensor(lengths).cuda(), torch.as_tensor(dense).cuda()
                eng.zero_grads()
                eng.mini_batch(dx, dl, dd, 0.9, 0.8, seed=i + 1, max_len=int(lengths.max()))
                eng.apply(1e-3, 1.0)
                losses.append(eng.loss.cpu().numpy().copy())
            eng.check()
            return np.stack(losses), eng.params.cpu().numpy().copy()
        finally:
            E._FUSED_CTC = old

    la, pa = steps(True)
    lb, pb = steps(False)
    assert np.all(np.isfinite(la))
    np.testing.assert_allclose(la, lb, rtol=1e-4, atol=1e-4)
    assert np.abs(pa - pb).max() < 1e-3 * np.abs(pb).max()      # (six Adam steps amplify the last bits of the first gradients)
