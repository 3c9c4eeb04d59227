"""Torch-tensor front of the C ABI: every function takes CUDA (ROCm) float32/int32
tensors, hands their device pointers and the current HIP stream to
libamdspeech.so, and returns tensors.  Torch is only the allocator / stream
provider here; there is no torch compute and no CPU fallback.
"""
import ctypes as C
import os

import torch

from . import lib as _l

MODE_MFCC, MODE_FBANK = 0, 1


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _chk_f32(*ts):
    for t in ts:
        if t is None:
            continue
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise ValueError("expected a contiguous float32 device tensor, got %s %s" % (t.dtype, t.device))


def _chk_f32_rows(*ts):
    """2-D float32 device matrices whose ROWS are contiguous (a column slice of a wider buffer is fine: ld = stride(0))."""
    for t in ts:
        if t is None:
            continue
        if not (t.is_cuda and t.dtype == torch.float32 and t.dim() == 2 and t.stride(1) == 1 and t.stride(0) >= t.shape[1]):
            raise ValueError("expected a float32 device matrix with contiguous rows, got %s %s %s" % (t.dtype, t.device, t.stride()))


def _chk_i32(*ts):
    for t in ts:
        if not (t.is_cuda and t.dtype == torch.int32 and t.is_contiguous()):
            raise ValueError("expected a contiguous int32 device tensor")


# ------------------------------------------------------------------ GEMM / Linear
def gemm(a, b, trans_a=False, trans_b=False, bias=None, out=None, accumulate=False):
    """C = op(A) @ op(B) (+ bias).  a is [M,K] (or [K,M] if trans_a), b [K,N] (or [N,K])."""
    _chk_f32(bias)
    _chk_f32_rows(a, b, out)
    M, K = (a.shape[1], a.shape[0]) if trans_a else a.shape
    K2, N = (b.shape[1], b.shape[0]) if trans_b else b.shape
    assert K == K2, (a.shape, b.shape)
    if out is None:
        out = torch.empty(M, N, device=a.device, dtype=torch.float32)
    _l.check(_l.load().amdspeech_gemm_f32(_stream(), int(trans_a), int(trans_b), M, N, K, _p(a), a.stride(0),
                                          _p(b), b.stride(0), _p(out), out.stride(0), _p(bias), int(accumulate)),
             "gemm_f32")
    return out


def gemm_bf16x3(a, b, trans_a=False, trans_b=False, bias=None, out=None, accumulate=False, single=False):
    """The same product as gemm() in split precision (bf16 hi/lo pairs, three bf16 MFMAs per product term, f32 accumulate);
    single=True: plain bf16 operands (one bf16 per value, one MFMA per term: precision = "bf16")."""
    _chk_f32(a, b, bias, out)
    M, K = (a.shape[1], a.shape[0]) if trans_a else a.shape
    K2, N = (b.shape[1], b.shape[0]) if trans_b else b.shape
    assert K == K2, (a.shape, b.shape)
    if out is None:
        out = torch.empty(M, N, device=a.device, dtype=torch.float32)
    fn = _l.load().amdspeech_gemm_bf16 if single else _l.load().amdspeech_gemm_bf16x3
    _l.check(fn(_stream(), int(trans_a), int(trans_b), M, N, K, _p(a), a.shape[1], _p(b), b.shape[1], _p(out), out.shape[1], _p(bias),
                int(accumulate)), "gemm_bf16" if single else "gemm_bf16x3")
    return out


def gemm_bf16(a, b, **kw):
    return gemm_bf16x3(a, b, single=True, **kw)


def gemm_bf16_packed(a, b, trans_a=False, trans_b=False, bias=None, out=None, accumulate=False):
    """The plain-bf16 product through bf16 COPIES of the operands (amdspeech_gemm_bf16_packed: what the H = 1024 LSTM path runs at
    precision = "bf16").  Returns None when the shape is not taken (the caller then uses gemm_bf16)."""
    _chk_f32(a, b, bias, out)
    M, K = (a.shape[1], a.shape[0]) if trans_a else a.shape
    K2, N = (b.shape[1], b.shape[0]) if trans_b else b.shape
    assert K == K2, (a.shape, b.shape)
    lib = _l.load()
    n = lib.amdspeech_gemm_bf16_packed_scratch_bytes(int(trans_a), int(trans_b), M, N, K, a.shape[1], b.shape[1])
    if n == 0:
        return None
    if out is None:
        out = torch.empty(M, N, device=a.device, dtype=torch.float32)
    scratch = torch.empty(n, device=a.device, dtype=torch.uint8)
    _l.check(lib.amdspeech_gemm_bf16_packed(_stream(), int(trans_a), int(trans_b), M, N, K, _p(a), a.shape[1], _p(b), b.shape[1], _p(out),
                                            out.shape[1], _p(bias), int(accumulate), _p(scratch), n), "gemm_bf16_packed")
    return out


def linear_fwd(x, w, b, out=None):
    """x [M,K] @ w [K,N] + b [N]."""
    _chk_f32(x, w, b, out)
    M, K = x.shape
    N = w.shape[1]
    if out is None:
        out = torch.empty(M, N, device=x.device, dtype=torch.float32)
    _l.check(_l.load().amdspeech_linear_fwd(_stream(), _p(x), _p(w), _p(b), _p(out), M, K, N), "linear_fwd")
    return out


def linear_bwd(x, w, dy, dw, db, need_dx=True, dx=None):
    """dw += x^T dy, db += colsum(dy); returns dx = dy w^T (or None)."""
    _chk_f32(x, w, dy, dw, db, dx)
    M, K = x.shape
    N = w.shape[1]
    if need_dx and dx is None:
        dx = torch.empty(M, K, device=x.device, dtype=torch.float32)
    _l.check(_l.load().amdspeech_linear_bwd(_stream(), _p(x), _p(w), _p(dy), _p(dx if need_dx else None),
                                            _p(dw), _p(db), M, K, N), "linear_bwd")
    return dx if need_dx else None


# -------------------------------------------------------------------- batch norm
def batchnorm_fwd(x, y, xhat, inv_std, eps=1e-3):
    """x, y [T,B,H] (may alias); xhat [T,B,H] or None; inv_std [T,H]."""
    _chk_f32(x, y, xhat, inv_std)
    T, B, H = x.shape
    _l.check(_l.load().amdspeech_batchnorm_fwd(_stream(), _p(x), _p(y), _p(xhat), _p(inv_std), T, B, H, eps),
             "batchnorm_fwd")


def batchnorm_bwd(dy, xhat, inv_std, dx):
    _chk_f32(dy, xhat, inv_std, dx)
    T, B, H = dy.shape
    _l.check(_l.load().amdspeech_batchnorm_bwd(_stream(), _p(dy), _p(xhat), _p(inv_std), _p(dx), T, B, H),
             "batchnorm_bwd")


def batchnorm_fwd_dp(x, y, xhat, inv_std, group, scratch, eps=1e-3):
    """Batch norm whose batch axis spans the ranks of `group` (dataparallel.Group): moments over the GLOBAL batch.
    scratch: float32 [2, T, H] device buffer."""
    _chk_f32(x, y, xhat, inv_std, scratch)
    T, B, H = x.shape
    lib, n = _l.load(), B * group.world
    gsum, gsq = scratch[0], scratch[1]
    _l.check(lib.amdspeech_batchnorm_sum(_stream(), _p(x), _p(None), n, _p(gsum), T, B, H), "batchnorm_sum")
    group.all_reduce_sum_(gsum)
    _l.check(lib.amdspeech_batchnorm_sum(_stream(), _p(x), _p(gsum), n, _p(gsq), T, B, H), "batchnorm_sum")
    group.all_reduce_sum_(gsq)
    _l.check(lib.amdspeech_batchnorm_apply(_stream(), _p(x), _p(gsum), _p(gsq), n, eps, _p(y), _p(xhat), _p(inv_std), T, B, H),
             "batchnorm_apply")


def batchnorm_bwd_dp(dy, xhat, inv_std, dx, group, scratch):
    _chk_f32(dy, xhat, inv_std, dx, scratch)
    T, B, H = dy.shape
    lib, n = _l.load(), B * group.world
    sums = scratch.view(-1)[:2 * T * H]
    _l.check(lib.amdspeech_batchnorm_bwd_sums(_stream(), _p(dy), _p(xhat), _p(sums), T, B, H), "batchnorm_bwd_sums")
    group.all_reduce_sum_(sums)
    _l.check(lib.amdspeech_batchnorm_bwd_apply(_stream(), _p(dy), _p(xhat), _p(inv_std), _p(sums), n, _p(dx), T, B, H),
             "batchnorm_bwd_apply")


# -------------------------------------------------------------------------- LSTM
class LstmWorkspace(object):
    """Owns the device workspace of one (T,B,H,L) LSTM stack and exposes the
    named regions as tensor views (no copies)."""

    def __init__(self, T, B, H, L, keep_in=1.0, keep_out=1.0, seed=0, device="cuda", precision=0, _share=None):
        self.lib = _l.load()
        self.desc = _l.LstmDesc(T, B, H, L, keep_in, keep_out, seed, int(precision))
        nbytes = self.lib.amdspeech_lstm_workspace_bytes(C.byref(self.desc))
        if nbytes == 0:
            raise _l.AmdSpeechError("lstm workspace: " + self.lib.amdspeech_last_error().decode())
        self.T, self.B, self.H, self.L = T, B, H, L
        if _share is None:
            self.buf = torch.empty(nbytes // 4, device=device, dtype=torch.float32)
        else:
            # (amdspeech_lstm_workspace_bytes of the owner covers every shorter run length of its shape; a view that does not fit would
            #  make the kernels write past the allocation -- an error in every build, not an assert)
            if _share.numel() * 4 < nbytes:
                raise _l.AmdSpeechError("lstm workspace: the layout for T = %d needs %d bytes, the shared allocation has %d"
                                        % (T, nbytes, _share.numel() * 4))
            self.buf = _share
        if self.buf.data_ptr() % 256 != 0:
            raise _l.AmdSpeechError("lstm workspace: allocation not 256-byte aligned")
        self.z0 = self._view(_l.WS_Z0, (T, B, H))
        self.ztop = self._view(_l.WS_ZTOP, (T, B, H))
        self.dztop = self._view(_l.WS_DZTOP, (T, B, H))
        self.dz0 = self._view(_l.WS_DZ0, (T, B, H))
        self._prefixes = {}
        self._root = self           # the owner of the allocation (prefix() views share it)
        self._armed = None          # (root only) {"fwd": (T, precision) | None, "bwd": ...}: layouts whose hand-off panels are prepared
        self._ever_armed = False    # (root only) the library keeps side-stream state (events) for this allocation
        self._fwd_seen = False      # (root only) lstm_fwd has run on this allocation: the next one may say AMDSPEECH_LSTM_SAME_WS
        self._lib_state = False     # (root only) some lstm call has run on it: the library may hold events for the allocation (released in __del__)

    def __del__(self):
        # the library's side stream may still be filling hand-off panels of this allocation (AMDSPEECH_LSTM_ARM_NEXT): order the
        # current stream behind that work before torch's caching allocator may hand the memory to someone else
        try:
            # (whenever it has EVER been armed: an eval forward in between clears `_armed`, the library's entry for the allocation
            #  -- its events, a possibly pending fill -- stays until released)
            if self._root is self and (self._ever_armed or self._fwd_seen or self._lib_state) and torch.cuda.is_available():
                self.lib.amdspeech_lstm_workspace_release(_stream(), _p(self.buf))
        except Exception:      # interpreter shutdown: nothing left to protect
            pass

    def prefix(self, T_run):
        """The same allocation laid out for a shorter sequence (the layout is a pure function of the
        descriptor, and everything is time-major, so a batch whose longest utterance has T_run < T frames
        runs T_run + L - 1 diagonals instead of T + L - 1 -- what tf.nn.dynamic_rnn's while-loop does with
        max(sequence_length), reference models/AcousticModel.py:276-278)."""
        if T_run >= self.T:
            return self
        ws = self._prefixes.get(T_run)
        if ws is None:
            if len(self._prefixes) > 64:
                self._prefixes.clear()
            ws = LstmWorkspace(T_run, self.B, self.H, self.L, device=self.buf.device,
                               precision=self.desc.precision, _share=self.buf)
            ws._root = self
            self._prefixes[T_run] = ws
        return ws

    def _offset(self, which):
        p = self.lib.amdspeech_lstm_ws_ptr(C.byref(self.desc), _p(self.buf), which)
        if not p:
            raise _l.AmdSpeechError("lstm_ws_ptr failed")
        return (p - self.buf.data_ptr()) // 4

    def _view(self, which, shape):
        off = self._offset(which)
        n = 1
        for s in shape:
            n *= s
        return self.buf[off:off + n].view(*shape)

    def set_dropout(self, keep_in, keep_out, seed):
        self.desc.keep_in, self.desc.keep_out, self.desc.seed = keep_in, keep_out, seed

    def final_state(self):
        """(h [L,B,H], c [L,B,H]) views of the state after the last frame."""
        T, B, H, L = self.T, self.B, self.H, self.L
        stride = (T + 1) * B * H
        oh, oc = self._offset(_l.WS_HFINAL), self._offset(_l.WS_CFINAL)
        h = torch.as_strided(self.buf, (L, B, H), (stride, H, 1), oh)
        c = torch.as_strided(self.buf, (L, B, H), (stride, H, 1), oc)
        return h, c


_ARM = os.environ.get("AMDSPEECH_ARM", "1") != "0"      # 0: every call fills its own hand-off panels


class CtcHead(object):
    """The CTC head fused into the whole-sequence LSTM kernels (amdspeech.h: amdspeech_ctc_head): output Linear + log-softmax +
    alpha follow the forward recurrence, beta + gradient + dlogits . W_o^T run ahead of the backward recurrence -- nothing of the
    CTC stage is left between the two launches.  Holds the tensors the two calls of a mini-batch share."""

    def __init__(self, w_out, b_out, logits, dense_labels, loss, dlogits, ctc_ws):
        _chk_f32(w_out, b_out, logits, loss, dlogits)
        _chk_i32(dense_labels)
        T, B, C_ = logits.shape
        if ctc_ws.shape[0] < T or tuple(ctc_ws.shape[1:]) != (B, C_, dense_labels.shape[1]):      # (a prefix of the frames it was sized for is fine)
            raise ValueError("CtcHead: the CTC workspace was sized for %r, the logits are %r with U = %d"
                             % (ctc_ws.shape, (T, B, C_), dense_labels.shape[1]))
        self.keep = (w_out, b_out, logits, dense_labels, loss, dlogits, ctc_ws)
        self.c = _l.CtcHead(_p(w_out).value, _p(b_out).value, _p(logits).value, _p(dense_labels).value, _p(loss).value,
                            _p(dlogits).value if dlogits is not None else None, _p(ctc_ws.buf).value, C_, dense_labels.shape[1])


def lstm_ctc_fusable(ws, C_, U, per_diagonal=False):
    """Whether lstm_fwd / lstm_bwd on this workspace layout take a CtcHead (amdspeech_lstm_ctc_fusable)."""
    if per_diagonal:
        return False
    return bool(ws.lib.amdspeech_lstm_ctc_fusable(C.byref(ws.desc), int(C_), int(U)))


def lstm_fwd(ws, kernels, kernel_stride, biases, bias_stride, lengths, h0=None, c0=None, training=False, per_diagonal=False, head=None):
    """kernels/biases: tensors whose data_ptr is layer 0's K / bias; strides in elements.
    training: lstm_bwd on the same workspace follows; the call then prepares that call's hand-off panels and the next forward
    call's (the other of the workspace's two sets) beside its kernel (amdspeech.h: AMDSPEECH_LSTM_ARM_NEXT), and the next calls
    of the same layout skip their fills.
"""
    _chk_i32(lengths)
    _chk_f32(h0, c0)
    root, key = ws._root, (ws.T, int(ws.desc.precision))
    if per_diagonal:                # the re-run of a mini-batch whose dataflow launch timed out (amdspeech.h): nothing is armed
        root._armed, training = None, False
    armed = _ARM and root._armed is not None and root._armed["fwd"] == key
    root._armed = None              # whatever runs now, the panels are in use
    inject, root._inject_timeout = getattr(root, "_inject_timeout", 0), 0       # (tests: ONE dataflow launch that gives up)
    # (SAME_WS: every view of one allocation shares B / H / L / precision, and nothing but the lstm calls writes into it)
    ws.desc.flags = ((_l.LSTM_ARMED if armed else 0) | (_l.LSTM_ARM_NEXT if (training and _ARM) else 0) |
                     (_l.LSTM_SAME_WS if (root._fwd_seen and _ARM) else 0) | (_l.LSTM_PER_DIAGONAL if per_diagonal else 0) |
                     (_l.LSTM_INJECT_TIMEOUT if inject else 0))
    root._fwd_seen = False          # (a call that raises leaves the history in an unknown state)
    try:
        if head is None:
            _l.check(ws.lib.amdspeech_lstm_fwd(_stream(), C.byref(ws.desc), _p(ws.buf), _p(kernels), kernel_stride,
                                               _p(biases), bias_stride, _p(lengths), _p(h0), _p(c0)), "lstm_fwd")
        else:
            _l.check(ws.lib.amdspeech_lstm_fwd_ctc(_stream(), C.byref(ws.desc), _p(ws.buf), _p(kernels), kernel_stride,
                                                   _p(biases), bias_stride, _p(lengths), _p(h0), _p(c0), C.byref(head.c)), "lstm_fwd_ctc")
        # (a launch-per-diagonal run at a shorter prefix writes over x-product history frames the NEXT whole-sequence launch would
        #  trust under SAME_WS: it leaves the history "unknown" -- the library keeps state for the allocation all the same)
        root._fwd_seen = not per_diagonal
        root._lib_state = True
    finally:
        ws.desc.flags = 0
    if training and _ARM:
        root._armed = {"fwd": key, "bwd": key}
        root._ever_armed = True


def lstm_pair_fusable(ws):
    """The layers of two stacks of this shape run side by side (amdspeech.h: amdspeech_lstm_pair_fusable)."""
    ws.desc.flags = 0
    return bool(ws.lib.amdspeech_lstm_pair_fusable(C.byref(ws.desc)))


def lstm_fwd_pair(ws_a, kernels_a, biases_a, ws_b, kernels_b, biases_b, kernel_stride, bias_stride, lengths, h0=None, c0=None):
    """Two stacks of one shape over one batch (a bidirectional model's two directions; h0 / c0: stack A's initial state).  The results
    of lstm_fwd(ws_a ...) followed by lstm_fwd(ws_b ...); where the library can, the two stacks' layers run side by side in one
    launch each (amdspeech.h: amdspeech_lstm_fwd_pair).  Nothing is armed: shapes that take this path have no hand-off panels."""
    _chk_i32(lengths)
    _chk_f32(h0, c0)
    for ws in (ws_a, ws_b):
        ws._root._armed, ws._root._fwd_seen, ws._root._lib_state = None, False, True
        ws.desc.flags = 0
    _l.check(ws_a.lib.amdspeech_lstm_fwd_pair(_stream(), C.byref(ws_a.desc), _p(ws_a.buf), _p(kernels_a), _p(biases_a),
                                              C.byref(ws_b.desc), _p(ws_b.buf), _p(kernels_b), _p(biases_b),
                                              kernel_stride, bias_stride, _p(lengths), _p(h0), _p(c0)), "lstm_fwd_pair")


def lstm_status(ws):
    """Synchronous check that no bounded wait of the persistent kernels timed out."""
    try:
        _l.check(ws.lib.amdspeech_lstm_status(C.byref(ws.desc), _p(ws.buf)), "lstm_status")
    except _l.AmdSpeechError:
        ws._root._armed = None          # (nothing of that launch's hand-off state is to be trusted: the next calls fill for themselves)
        ws._root._fwd_seen = False
        raise


def lstm_beside_forward(ws, stream):
    """Orders `stream` (a torch.cuda.Stream) behind the point just in front of the last lstm_fwd launch on `ws` and returns the
    number of XCDs that launch leaves idle (0: not a whole-sequence dataflow launch -- nothing is ordered).  Work-queue kernels
    enqueued on `stream` afterwards do their work beside the forward recurrence (amdspeech.h: amdspeech_lstm_beside_forward)."""
    rc = _l.load().amdspeech_lstm_beside_forward(C.c_void_p(stream.cuda_stream), _p(ws._root.buf))
    if rc < 0:
        _l.check(rc, "lstm_beside_forward")
    return rc


def lstm_beside_tail(ws, stream):
    """Orders `stream` behind the last lstm_bwd's whole-sequence kernel on `ws`, in FRONT of the weight-gradient launches that follow
    it on the caller's stream (amdspeech.h: amdspeech_lstm_beside_tail).  Returns 0 (nothing ordered) or flags: 1 ordered, 2 dZ_0 is
    complete at that point."""
    rc = _l.load().amdspeech_lstm_beside_tail(C.c_void_p(stream.cuda_stream), _p(ws._root.buf))
    if rc < 0:
        _l.check(rc, "lstm_beside_tail")
    return rc


def lstm_bwd(ws, kernels, kernel_stride, dkernels, dbiases, bias_stride, lengths, per_diagonal=False, head=None):
    _chk_i32(lengths)
    root, key = ws._root, (ws.T, int(ws.desc.precision))
    armed = root._armed is not None and root._armed["bwd"] == key and not per_diagonal
    if root._armed is not None:
        root._armed["bwd"] = None       # (used once; the forward half stays valid for the next lstm_fwd)
    root._lib_state = True
    inject, root._inject_timeout_bwd = getattr(root, "_inject_timeout_bwd", 0), 0      # (tests: ONE backward dataflow launch that gives up)
    ws.desc.flags = ((_l.LSTM_ARMED if armed else 0) | (_l.LSTM_PER_DIAGONAL if per_diagonal else 0) |
                     (_l.LSTM_INJECT_TIMEOUT if inject else 0))
    try:
        if head is None:
            _l.check(ws.lib.amdspeech_lstm_bwd(_stream(), C.byref(ws.desc), _p(ws.buf), _p(kernels), kernel_stride,
                                               _p(dkernels), _p(dbiases), bias_stride, _p(lengths)), "lstm_bwd")
        else:
            _l.check(ws.lib.amdspeech_lstm_bwd_ctc(_stream(), C.byref(ws.desc), _p(ws.buf), _p(kernels), kernel_stride,
                                                   _p(dkernels), _p(dbiases), bias_stride, _p(lengths), C.byref(head.c)), "lstm_bwd_ctc")
    finally:
        ws.desc.flags = 0


def lstm_bwd_pair(ws_a, kernels_a, dkernels_a, dbiases_a, ws_b, kernels_b, dkernels_b, dbiases_b, kernel_stride, bias_stride, lengths):
    """The backward passes of the two stacks of lstm_fwd_pair (dztop of either workspace filled): the results of two lstm_bwd calls,
    side by side where the library can (amdspeech.h: amdspeech_lstm_bwd_pair)."""
    _chk_i32(lengths)
    for ws in (ws_a, ws_b):
        if ws._root._armed is not None:
            ws._root._armed["bwd"] = None
        ws._root._lib_state = True
        ws.desc.flags = 0
    _l.check(ws_a.lib.amdspeech_lstm_bwd_pair(_stream(), C.byref(ws_a.desc), _p(ws_a.buf), _p(kernels_a), _p(dkernels_a), _p(dbiases_a),
                                              C.byref(ws_b.desc), _p(ws_b.buf), _p(kernels_b), _p(dkernels_b), _p(dbiases_b),
                                              kernel_stride, bias_stride, _p(lengths)), "lstm_bwd_pair")


def lstm_dropout_multipliers(ws, which, layer):
    """[T,B,H] inverted-dropout multipliers (mask / keep) the LSTM calls on `ws` apply with its current keep_in / keep_out /
    seed: which = "in" / "out" mask of `layer` (DropoutWrapper, reference :227-233)."""
    out = torch.empty(ws.T, ws.B, ws.H, device=ws.buf.device, dtype=torch.float32)
    _l.check(ws.lib.amdspeech_lstm_dropout_multipliers(_stream(), C.byref(ws.desc), {"in": 0, "out": 1}[which], int(layer),
                                                       _p(out)), "lstm_dropout_multipliers")
    return out


def reverse_sequences(x, lengths, out=None, accumulate=False):
    """Time-major [T,B,H]: out[t,b] = x[len_b-1-t, b] for t < len_b, 0 beyond (tf.reverse_sequence; self-adjoint)."""
    _chk_f32(x, out)
    _chk_i32(lengths)
    T, B, H = x.shape
    if out is None:
        out = torch.empty_like(x)
    _l.check(_l.load().amdspeech_reverse_sequences(_stream(), _p(x), _p(out), _p(lengths), T, B, H, int(bool(accumulate))),
             "reverse_sequences")
    return out


# --------------------------------------------------------------------------- CTC
class CtcWorkspace(object):
    def __init__(self, T, B, C_, U, device="cuda"):
        self.lib = _l.load()
        n = self.lib.amdspeech_ctc_workspace_bytes(T, B, C_, U)
        if n == 0:
            raise _l.AmdSpeechError("ctc workspace: bad shape")
        self.shape = (T, B, C_, U)
        self.buf = torch.empty(n, device=device, dtype=torch.uint8)
        self.greedy_ws = torch.empty(T * B, device=device, dtype=torch.int32)


def ctc_loss_fwd_bwd(logits, dense_labels, lengths, ws=None, loss=None, dlogits=None, stage=0):
    """logits [T,B,C]; dense_labels int32 [B,U] (0-padded, reference labels_ph);
    returns (loss [B], dlogits [T,B,C]).  stage 1 / 2: the two halves of the call (amdspeech.h), same arguments."""
    _chk_f32(logits, loss, dlogits)
    _chk_i32(dense_labels, lengths)
    T, B, C_ = logits.shape
    U = dense_labels.shape[1]
    if ws is None or ws.shape[1:] != (B, C_, U) or ws.shape[0] < T:     # a longer-T workspace serves a prefix
        ws = CtcWorkspace(T, B, C_, U, logits.device)
    if loss is None:
        loss = torch.empty(B, device=logits.device, dtype=torch.float32)
    if dlogits is None:
        dlogits = torch.empty_like(logits)
    _l.check(ws.lib.amdspeech_ctc_loss_fwd_bwd_staged(_stream(), _p(logits), _p(dense_labels), _p(lengths), T, B, C_, U,
                                                      _p(loss), _p(dlogits), _p(ws.buf), int(stage)), "ctc_loss_fwd_bwd")
    return loss, dlogits


def ctc_greedy_decode(logits, lengths, ws=None):
    """Returns (ids int32 [B,T] padded with C, out_len int32 [B])."""
    _chk_f32(logits)
    _chk_i32(lengths)
    T, B, C_ = logits.shape
    scratch = ws.greedy_ws if ws is not None and ws.shape[:2] == (T, B) else \
        torch.empty(T * B, device=logits.device, dtype=torch.int32)
    ids = torch.empty(B, T, device=logits.device, dtype=torch.int32)
    out_len = torch.empty(B, device=logits.device, dtype=torch.int32)
    _l.check(_l.load().amdspeech_ctc_greedy_decode(_stream(), _p(logits), _p(lengths), T, B, C_, _p(ids),
                                                   _p(out_len), _p(scratch)), "ctc_greedy_decode")
    return ids, out_len


def merge_repeated(ids, lens, pad):
    """In place on the device: collapse consecutive duplicate labels of each row."""
    _chk_i32(ids, lens)
    B, T = ids.shape
    _l.check(_l.load().amdspeech_merge_repeated(_stream(), _p(ids), _p(lens), T, B, int(pad)), "merge_repeated")
    return ids, lens


def edit_distance(a, a_len, b, b_len):
    """Levenshtein distance per row pair, int32 [n] on the device."""
    _chk_i32(a, a_len, b, b_len)
    n = a.shape[0]
    out = torch.empty(n, device=a.device, dtype=torch.int32)
    _l.check(_l.load().amdspeech_edit_distance(_stream(), _p(a), _p(a_len), a.shape[1], _p(b), _p(b_len),
                                               b.shape[1], n, _p(out)), "edit_distance")
    return out


def edit_distance_host(a, a_len, b, b_len):
    """Levenshtein distance per row pair on the HOST (numpy int32 in, int32 [n] out): for predictions decoded on the host."""
    import numpy as np
    a = np.ascontiguousarray(a, np.int32); b = np.ascontiguousarray(b, np.int32)
    a_len = np.ascontiguousarray(a_len, np.int32); b_len = np.ascontiguousarray(b_len, np.int32)
    n = a.shape[0]
    out = np.empty(n, np.int32)
    _l.check(_l.load().amdspeech_edit_distance_host(a.ctypes.data_as(C.c_void_p), a_len.ctypes.data_as(C.c_void_p), a.shape[1],
                                                    b.ctypes.data_as(C.c_void_p), b_len.ctypes.data_as(C.c_void_p), b.shape[1], n,
                                                    out.ctypes.data_as(C.c_void_p)), "edit_distance_host")
    return out


def ctc_beam_search(logits, lengths, beam_width=100, merge_repeated=True, max_threads=0):
    """Host-side prefix beam search (evaluation path).  logits: [T,B,C] tensor or array (copied to the
    host), lengths: ints.  Returns (ids int32 [B,T] numpy padded with C, out_len [B], log_prob [B]).
    max_threads > 0 caps the decode threads of the call (default: one per utterance)."""
    import numpy as np
    host = logits.detach().cpu().numpy() if torch.is_tensor(logits) else np.asarray(logits)
    host = np.ascontiguousarray(host, np.float32)
    T, B, C_ = host.shape
    lens = np.ascontiguousarray(lengths.cpu().numpy() if torch.is_tensor(lengths) else lengths, np.int32)
    ids = np.empty((B, T), np.int32)
    out_len = np.empty(B, np.int32)
    logp = np.empty(B, np.float32)
    _l.check(_l.load().amdspeech_ctc_beam_search_host_mt(
        host.ctypes.data_as(C.c_void_p), lens.ctypes.data_as(C.c_void_p), T, B, C_, int(beam_width),
        int(bool(merge_repeated)), ids.ctypes.data_as(C.c_void_p), out_len.ctypes.data_as(C.c_void_p),
        logp.ctypes.data_as(C.c_void_p), int(max_threads)), "ctc_beam_search_host")
    return ids, out_len, logp


# --------------------------------------------------------------------- optimiser
_optim_ws = {}


def clip_adam(params, grads, m, v, clip, lr_t, beta1=0.9, beta2=0.999, eps=1e-8, norm_out=None):
    _chk_f32(params, grads, m, v)
    n = params.numel()
    key = params.device
    if key not in _optim_ws:
        _optim_ws[key] = torch.empty(_l.load().amdspeech_optim_workspace_bytes(n) // 4, device=params.device,
                                     dtype=torch.float32)
    if norm_out is None:
        norm_out = torch.empty(1, device=params.device, dtype=torch.float32)
    _l.check(_l.load().amdspeech_clip_adam(_stream(), _p(params), _p(grads), _p(m), _p(v), n, clip, lr_t, beta1,
                                           beta2, eps, _p(norm_out), _p(_optim_ws[key])), "clip_adam")
    return norm_out


# --------------------------------------------------------------------- front end
_frontend_ws = {}


def audio_probe(path):
    """(sample_rate, channels, frames) from the header of a WAVE / FLAC / NIST SPHERE file (host)."""
    sr, ch, fr = C.c_int(), C.c_int(), C.c_long()
    _l.check(_l.load().amdspeech_audio_probe(str(path).encode(), C.byref(sr), C.byref(ch), C.byref(fr)), "audio_probe")
    return sr.value, ch.value, fr.value


def audio_decode(path, verify=False):
    """Mono float32 numpy array in [-1, 1) and the file's sample rate (host; ctypes releases the GIL, so
    a thread pool decodes files in parallel).  verify: also check the FLAC STREAMINFO MD5."""
    import numpy as np
    lib = _l.load()
    enc = str(path).encode()
    sr, ch, fr = C.c_int(), C.c_int(), C.c_long()
    _l.check(lib.amdspeech_audio_probe(enc, C.byref(sr), C.byref(ch), C.byref(fr)), "audio_probe")
    out = np.empty(max(fr.value, 1), np.float32)
    got = C.c_long()
    _l.check(lib.amdspeech_audio_decode(enc, C.c_void_p(out.ctypes.data), out.size, C.byref(got), C.byref(sr),
                                        int(bool(verify))), "audio_decode")
    return out[:got.value], sr.value


_resample_ws = {}


def resample(pcm, n_samples, rate_in, rate_out):
    """pcm float32 [B, n_max] (device), n_samples python ints -> (out [B, out_max] device, new lengths):
    librosa.load's resampling step (resampy kaiser_best semantics), on the GPU."""
    _chk_f32(pcm)
    lib = _l.load()
    B, n_max = pcm.shape
    n_out = [lib.amdspeech_resample_num_samples(int(n), int(rate_in), int(rate_out)) for n in n_samples]
    out_max = max(max(n_out), 1)
    key = (B, pcm.device)
    ws = _resample_ws.get(key)
    if ws is None:
        if len(_resample_ws) > 8:
            _resample_ws.clear()
        ws = _resample_ws[key] = torch.empty(lib.amdspeech_resample_workspace_bytes(B), device=pcm.device,
                                             dtype=torch.uint8)
    out = torch.empty(B, out_max, device=pcm.device, dtype=torch.float32)
    ns = (C.c_int * B)(*[int(v) for v in n_samples])
    _l.check(lib.amdspeech_resample(_stream(), _p(pcm), ns, B, n_max, int(rate_in), int(rate_out), _p(out), out_max,
                                    _p(ws)), "resample")
    return out, n_out


def frontend(pcm, n_samples, sample_rate, mode, t_max, n_mfcc=20):
    """pcm float32 [B, n_max] (device), n_samples: python ints.  Returns
    (feat [t_max, B, D] device, n_frames list of UNtruncated frame counts)."""
    _chk_f32(pcm)
    lib = _l.load()
    B, n_max = pcm.shape
    imode = MODE_MFCC if mode == "mfcc" else MODE_FBANK
    D = n_mfcc if imode == MODE_MFCC else 120
    key = (imode, B, n_max, sample_rate, pcm.device)
    ws = _frontend_ws.get(key)
    if ws is None:
        nbytes = lib.amdspeech_frontend_workspace_bytes(imode, B, n_max, sample_rate)
        if nbytes == 0:
            raise _l.AmdSpeechError("frontend workspace: bad arguments")
        if len(_frontend_ws) > 8:
            _frontend_ws.clear()
        ws = _frontend_ws[key] = torch.empty(nbytes, device=pcm.device, dtype=torch.uint8)
    feat = torch.empty(t_max, B, D, device=pcm.device, dtype=torch.float32)
    ns = (C.c_int * B)(*[int(v) for v in n_samples])
    nf = (C.c_int * B)()
    if imode == MODE_MFCC:
        rc = lib.amdspeech_frontend_mfcc(_stream(), _p(pcm), ns, B, n_max, sample_rate, n_mfcc, t_max, _p(feat),
                                         nf, _p(ws))
    else:
        rc = lib.amdspeech_frontend_fbank(_stream(), _p(pcm), ns, B, n_max, sample_rate, t_max, _p(feat), nf,
                                          _p(ws))
    _l.check(rc, "frontend_" + mode)
    return feat, list(nf)          # stream-ordered; the cached workspace outlives the kernels
