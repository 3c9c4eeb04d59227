"""amdspeech_lstm_fwd_pair (include/amdspeech.h): two stacks of one shape over one batch -- a bidirectional model's two directions
(BASELINE configs[4]; the reference itself builds a unidirectional dynamic_rnn, models/AcousticModel.py:266-297) -- give the results
of two amdspeech_lstm_fwd calls; at 1024 units in plain bf16 their layers run side by side in one launch each (lstm_fwd_big1 / lstm_bwd_big1: one XCD per batch tile)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

_BIG1 = os.environ.get("AMDSPEECH_BIG1", "1") != "0"      # (the switch test below runs this file again with AMDSPEECH_BIG1=0)


def _stack(T, B, H, L, precision, seed):
    from rnn_speech_amd import ops
    g = torch.Generator(device="cpu").manual_seed(seed)
    ws = ops.LstmWorkspace(T, B, H, L, precision=precision)
    k = (torch.randn(L, 2 * H, 4 * H, generator=g) * (0.6 / np.sqrt(H))).cuda()
    b = (torch.randn(L, 4 * H, generator=g) * 0.1).cuda()
    z0 = torch.randn(T, B, H, generator=g).cuda()
    return ws, k, b, z0


def _run_two_calls(ws, k, b, z0, lengths, h0, c0):
    from rnn_speech_amd import ops
    ws.z0.copy_(z0)
    ops.lstm_fwd(ws, k, k.stride(0), b, b.stride(0), lengths, h0, c0)
    ops.lstm_status(ws)
    h, c = ws.final_state()
    return ws.ztop.clone(), h.clone(), c.clone()


@pytest.mark.parametrize("T,B,H,L,precision,side_by_side", [
    (37, 64, 1024, 2, 2, True),        # side by side: all four batch tiles of both stacks
    (21, 40, 1024, 3, 2, True),        # ragged third batch tile, the fourth XCD pair idle
    (9, 7, 1024, 1, 2, True),          # one batch tile
    (12, 40, 1024, 2, 1, False),       # bf16x3: the XCD pairs, one stack after the other
    (30, 20, 128, 2, 0, False),        # whole-sequence kernels: two calls
    (10, 5, 64, 2, 0, False),          # launch-per-diagonal kernels: two calls
])
def test_pair_is_the_two_calls(T, B, H, L, precision, side_by_side):
    from rnn_speech_amd import ops
    wa, ka, ba, za = _stack(T, B, H, L, precision, 1)
    wb, kb, bb, zb = _stack(T, B, H, L, precision, 2)
    side_by_side = side_by_side and _BIG1
    assert ops.lstm_pair_fusable(wa) == side_by_side
    rng = np.random.RandomState(T)
    lengths = rng.randint(1, T + 1, size=B).astype(np.int32)
    lengths[0] = T
    if B > 2:
        lengths[2] = 0
    lengths = torch.from_numpy(lengths).cuda()
    h0 = (torch.randn(L, B, H) * 0.3).cuda()
    c0 = (torch.randn(L, B, H) * 0.3).cuda()
    want_a = _run_two_calls(wa, ka, ba, za, lengths, h0, c0)
    want_b = _run_two_calls(wb, kb, bb, zb, lengths, None, None)
    for ws in (wa, wb):
        ws.buf.zero_()          # (whatever the pair leaves out would show)
    wa.z0.copy_(za)
    wb.z0.copy_(zb)
    ops.lstm_fwd_pair(wa, ka, ba, wb, kb, bb, ka.stride(0), ba.stride(0), lengths, h0, c0)
    ops.lstm_status(wa)
    ops.lstm_status(wb)
    for ws, want in ((wa, want_a), (wb, want_b)):
        h, c = ws.final_state()
        for got, ref in zip((ws.ztop, h, c), want):
            # same products in the same order on either kernel (K slices of 128 rows per wave, partial sums added wave by wave)
            assert torch.equal(got, ref), float((got - ref).abs().max())


def test_pair_with_dropout_keeps_the_stacks_streams_apart():
    from rnn_speech_amd import ops
    T, B, H, L = 16, 64, 1024, 2
    wa, ka, ba, za = _stack(T, B, H, L, 2, 3)
    wb, kb, bb, zb = _stack(T, B, H, L, 2, 4)
    lengths = torch.full((B,), T, dtype=torch.int32).cuda()
    wa.set_dropout(0.9, 0.8, 11)
    wb.set_dropout(0.9, 0.8, 12)
    want_a = _run_two_calls(wa, ka, ba, za, lengths, None, None)
    want_b = _run_two_calls(wb, kb, bb, zb, lengths, None, None)
    wa.z0.copy_(za)
    wb.z0.copy_(zb)
    ops.lstm_fwd_pair(wa, ka, ba, wb, kb, bb, ka.stride(0), ba.stride(0), lengths)
    ops.lstm_status(wa)
    assert torch.equal(wa.ztop, want_a[0]) and torch.equal(wb.ztop, want_b[0])
    assert not torch.equal(wa.ztop != 0, wb.ztop != 0)


def _bwd_two_calls(ws, k, lengths, dztop):
    from rnn_speech_amd import ops
    dk, db = torch.zeros_like(k), torch.zeros(k.shape[0], k.shape[2], device="cuda")
    ws.dztop.copy_(dztop)
    ops.lstm_bwd(ws, k, k.stride(0), dk, db, db.stride(0), lengths)
    ops.lstm_status(ws)
    return dk, db, ws.dz0.clone()


@pytest.mark.parametrize("T,B,H,L,precision,side_by_side,keep", [
    (37, 64, 1024, 2, 2, True, 1.0),        # side by side: all four batch tiles of both stacks
    (64, 40, 1024, 3, 2, True, 0.8),        # ragged third batch tile, dropout
    (64, 7, 1024, 1, 2, True, 1.0),         # one batch tile
    (12, 40, 1024, 2, 1, False, 1.0),       # bf16x3: the XCD pairs, one stack after the other
    (30, 20, 128, 2, 0, False, 0.8),        # whole-sequence kernels: two calls
])
def test_backward_pair_is_the_two_calls(T, B, H, L, precision, side_by_side, keep):
    from rnn_speech_amd import ops
    side_by_side = side_by_side and _BIG1
    wa, ka, ba, za = _stack(T, B, H, L, precision, 5)
    wb, kb, bb, zb = _stack(T, B, H, L, precision, 6)
    rng = np.random.RandomState(T + B)
    lengths = rng.randint(1, T + 1, size=B).astype(np.int32)
    lengths[0] = T
    if B > 2:
        lengths[2] = 0
    lengths = torch.from_numpy(lengths).cuda()
    wa.set_dropout(keep, keep, 21)
    wb.set_dropout(keep, keep, 22)
    da, dbt = torch.randn(T, B, H).cuda() * 0.1, torch.randn(T, B, H).cuda() * 0.1
    _run_two_calls(wa, ka, ba, za, lengths, None, None)
    _run_two_calls(wb, kb, bb, zb, lengths, None, None)
    want_a = _bwd_two_calls(wa, ka, lengths, da)
    want_b = _bwd_two_calls(wb, kb, lengths, dbt)
    for ws in (wa, wb):
        ws.buf.zero_()
    wa.z0.copy_(za)
    wb.z0.copy_(zb)
    ops.lstm_fwd_pair(wa, ka, ba, wb, kb, bb, ka.stride(0), ba.stride(0), lengths)
    wa.dztop.copy_(da)
    wb.dztop.copy_(dbt)
    dka, dba, dkb, dbb = torch.zeros_like(ka), torch.zeros_like(ba), torch.zeros_like(kb), torch.zeros_like(bb)
    ops.lstm_bwd_pair(wa, ka, dka, dba, wb, kb, dkb, dbb, ka.stride(0), dba.stride(0), lengths)
    ops.lstm_status(wa)
    ops.lstm_status(wb)
    for got, want in (((dka, dba, wa.dz0), want_a), ((dkb, dbb, wb.dz0), want_b)):
        for g, r, name in zip(got, want, ("dK", "db", "dZ0")):
            if side_by_side:
                # the same bf16 products; the partial sums of dh are grouped by other pairs of unit blocks: f32 rounding, which moves a few
                # values of dG across a bf16 rounding boundary (the accuracy itself: tests/test_gpu_fullsize_cfg3.py against the oracle)
                err = float((g - r).abs().max()) / (float(r.abs().max()) + 1e-30)
                assert err < 3e-3, (name, err)
            else:      # the same launches in the same order (split-K sums of the batched products: not bit-stable from run to run)
                assert torch.allclose(g, r, rtol=1e-4, atol=1e-5 * float(r.abs().max())), name


def test_engine_bidirectional_bf16_pair_against_two_calls():
    """Engine(bidirectional=True, precision="bf16") at 1024 units makes ONE call per pass for the two directions (ops.lstm_fwd_pair /
    lstm_bwd_pair); AMDSPEECH_BIDIR_PAIR=0 keeps the two calls per pass of rounds 3 - 4: same logits bit for bit, the
    same gradients up to the summation order of the backward partial tiles."""
    from rnn_speech_amd import engine as eng_mod
    from rnn_speech_amd.engine import Engine
    L, H, D, C, B, T, U = 2, 1024, 40, 80, 64, 32, 8
    rng = np.random.RandomState(3)
    x = torch.from_numpy(rng.randn(T, B, D).astype(np.float32)).cuda()
    lengths = rng.randint(T // 2, T + 1, size=B).astype(np.int32)
    dense = np.zeros((B, U), np.int32)
    for b in range(B):
        n = rng.randint(1, min(U - 1, int(lengths[b]) // 3 + 1))
        dense[b, :n] = rng.randint(1, C - 1, size=n)
        dense[b, n] = C - 1
    lengths, dense = torch.from_numpy(lengths).cuda(), torch.from_numpy(dense).cuda()
    results = []
    for pair in (True, False):
        old = eng_mod._BIDIR_PAIR
        eng_mod._BIDIR_PAIR = pair
        try:
            eng = Engine(L, H, D, C, B, T, U, seed=11, precision="bf16", bidirectional=True)
            eng.zero_grads()
            eng.mini_batch(x, lengths, dense, keep_in=0.9, keep_out=0.9, seed=5)
            torch.cuda.synchronize()
            assert eng.healthy()
            results.append((eng.logits.clone(), eng.loss.clone(), eng.grads.clone()))
        finally:
            eng_mod._BIDIR_PAIR = old
    (lg_a, loss_a, g_a), (lg_b, loss_b, g_b) = results
    if _BIG1:
        assert torch.equal(lg_a, lg_b) and torch.equal(loss_a, loss_b)
    else:
        assert torch.allclose(lg_a, lg_b, rtol=1e-4, atol=1e-5)
    assert float((g_a - g_b).abs().max()) < 3e-3 * float(g_b.abs().max())
    assert float(g_b.abs().max()) > 0


def test_two_calls_switch_keeps_parity():
    """AMDSPEECH_BIG1=0: no shape runs two stacks side by side -- amdspeech_lstm_pair_fusable answers 0 and the pair
    entry points make the two calls on the XCD-pair kernels.  This file again, in a child process (the library reads the switch once)."""
    import subprocess
    import sys
    if not _BIG1:
        pytest.skip("already the child")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-q", "-x"],
                         env=dict(os.environ, AMDSPEECH_BIG1="0"), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]


@pytest.mark.parametrize("T,B,L", [(32, 64, 2), (64, 40, 2), (64, 7, 1)])
def test_one_xcd_kernels_against_exact_f32(T, B, L):
    """lstm_fwd_big1 / lstm_bwd_big1 (precision 2, T*B a multiple of 64: the bf16 operand copies) against the exact-f32 kernels of the
    same shape on the same inputs: bf16-sized differences only -- a wrong unit block, K slice or batch tile would be O(1).  (The accuracy
    itself, against the float64 oracle over 998 frames: tests/test_gpu_fullsize_cfg3.py.)"""
    from rnn_speech_amd import ops
    H = 1024
    rng = np.random.RandomState(B)
    lengths = rng.randint(T // 2, T + 1, size=B).astype(np.int32)
    lengths[0] = T
    lengths = torch.from_numpy(lengths).cuda()
    dz = torch.randn(T, B, H).cuda() * 0.1
    out = {}
    for precision in (0, 2):
        wa, ka, ba, za = _stack(T, B, H, L, precision, 7)
        wb, kb, bb, zb = _stack(T, B, H, L, precision, 8)
        wa.z0.copy_(za)
        wb.z0.copy_(zb)
        ops.lstm_fwd_pair(wa, ka, ba, wb, kb, bb, ka.stride(0), ba.stride(0), lengths)
        wa.dztop.copy_(dz)
        wb.dztop.copy_(dz)
        dka, dba, dkb, dbb = torch.zeros_like(ka), torch.zeros_like(ba), torch.zeros_like(kb), torch.zeros_like(bb)
        ops.lstm_bwd_pair(wa, ka, dka, dba, wb, kb, dkb, dbb, ka.stride(0), dba.stride(0), lengths)
        ops.lstm_status(wa)
        ops.lstm_status(wb)
        out[precision] = [t.clone() for t in (wa.ztop, wb.ztop, dka, dkb, dba, dbb, wa.dz0, wb.dz0)]
    for got, ref, name in zip(out[2], out[0], ("ztop_a", "ztop_b", "dK_a", "dK_b", "db_a", "db_b", "dZ0_a", "dZ0_b")):
        err = float((got - ref).abs().max()) / float(ref.abs().max())
        assert err < 3e-2, (name, err)
        assert float(ref.abs().max()) > 0, name


@pytest.mark.parametrize("precision,bidirectional,tol", [("f32", False, 2e-4), ("bf16", False, 4e-2), ("bf16", True, 4e-2)])
def test_per_diagonal_rerun_at_1024_units_leaves_the_per_layer_kernels(precision, bidirectional, tol):
    """AMDSPEECH_LSTM_PER_DIAGONAL (the repeat of a mini-batch whose launch gave up waiting, include/amdspeech.h) takes no kernel with
    bounded waits at 1024 units either: not lstm_fwd_big / lstm_bwd_big, not the one-XCD kernels, not the pair entry points -- same
    results as the default kernels (f32: summation order; bf16: the launch-per-diagonal kernels multiply in bf16x3, a superset)."""
    from rnn_speech_amd.engine import Engine
    L, H, D, C, B, T, U = 2, 1024, 40, 80, 64, 16, 6
    rng = np.random.RandomState(5)
    x = torch.from_numpy(rng.randn(T, B, D).astype(np.float32)).cuda()
    lengths = rng.randint(T // 2, T + 1, size=B).astype(np.int32)
    dense = np.zeros((B, U), np.int32)
    for b in range(B):
        n = rng.randint(1, min(U - 1, int(lengths[b]) // 3 + 1))
        dense[b, :n] = rng.randint(1, C - 1, size=n)
        dense[b, n] = C - 1
    lengths, dense = torch.from_numpy(lengths).cuda(), torch.from_numpy(dense).cuda()
    eng = Engine(L, H, D, C, B, T, U, seed=13, precision=precision, bidirectional=bidirectional)
    out = []
    for per_diagonal in (False, True):
        eng.zero_grads()
        eng.mini_batch(x, lengths, dense, per_diagonal=per_diagonal)
        torch.cuda.synchronize()
        assert eng.healthy()
        out.append((eng.logits.clone(), eng.loss.clone(), eng.grads.clone()))
    (lg_a, loss_a, g_a), (lg_b, loss_b, g_b) = out
    assert float((lg_a - lg_b).abs().max()) < tol * float(lg_a.abs().max())
    assert float((g_a - g_b).abs().max()) < tol * float(g_a.abs().max())
    assert torch.allclose(loss_a, loss_b, rtol=max(tol, 1e-3), atol=1e-2)
