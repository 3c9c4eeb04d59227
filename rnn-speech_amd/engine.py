"""Device-side training engine for the acoustic model: flat parameter / gradient /
Adam buffers in HBM, the LSTM + CTC workspaces, and the op sequence of one
mini-batch and one optimiser step.  Everything numeric is a libamdspeech call
(ops.py); torch provides memory, streams and (for data parallel) the RCCL
all-reduce.

Mirrors what one `session.run` does in the reference:
  mini-batch  = models/AcousticModel.py:634-660  (forward, CTC, gradients += )
  apply       = models/AcousticModel.py:672-703  (clip_by_global_norm + Adam)
"""
import contextlib
import math
import os

import numpy as np
import torch

from . import lib as _lib
from . import ops


def _pad64(n):
    return (n + 63) // 64 * 64


class ParamLayout(object):
    """Flat fp32 layout [input_w | input_b | (kernel_l | bias_l)*L | output_w | output_b],
    every tensor starting on a 256-byte boundary (pads are zero and stay zero)."""

    def __init__(self, num_layers, hidden, input_dim, num_labels, bidirectional=False):
        self.L, self.H, self.D, self.C = num_layers, hidden, input_dim, num_labels
        self.bidirectional = bool(bidirectional)
        off = 0
        self.slots = {}

        def take(name, shape):
            nonlocal off
            n = 1
            for s in shape:
                n *= s
            self.slots[name] = (off, shape)
            off += _pad64(n)

        take("input_w", (input_dim, hidden))
        take("input_b", (hidden,))
        for l in range(num_layers):
            take("kernel_%d" % l, (2 * hidden, 4 * hidden))
            take("bias_%d" % l, (4 * hidden,))
        if self.bidirectional:           # the backward-direction stack: same shapes, same stride between layers
            for l in range(num_layers):
                take("bw_kernel_%d" % l, (2 * hidden, 4 * hidden))
                take("bw_bias_%d" % l, (4 * hidden,))
        take("output_w", ((2 if self.bidirectional else 1) * hidden, num_labels))
        take("output_b", (num_labels,))
        self.total = off
        if num_layers > 1:
            self.kernel_stride = self.slots["kernel_1"][0] - self.slots["kernel_0"][0]
            self.bias_stride = self.slots["bias_1"][0] - self.slots["bias_0"][0]
        else:
            self.kernel_stride = self.bias_stride = 0

    def names(self):
        return list(self.slots.keys())

    def view(self, flat, name):
        off, shape = self.slots[name]
        n = 1
        for s in shape:
            n *= s
        return flat[off:off + n].view(*shape)

    def num_params(self):
        tot = 0
        for _, shape in self.slots.values():
            n = 1
            for s in shape:
                n *= s
            tot += n
        return tot


_BESIDE_FORWARD = os.environ.get("AMDSPEECH_BESIDE_FORWARD", "1") != "0"      # 0: the side work always goes beside the CTC stage
_BESIDE_TAIL = os.environ.get("AMDSPEECH_BESIDE_TAIL", "1") != "0"            # 0: the dense layers' weight gradients behind the LSTM's
_BIDIR_PAIR = os.environ.get("AMDSPEECH_BIDIR_PAIR", "1") != "0"              # 0: a bidirectional model's two stacks as two lstm_fwd calls
_FUSED_CTC = os.environ.get("AMDSPEECH_FUSED_CTC", "1") != "0"                # 0: the CTC stage as launches between the two recurrence kernels


# amdspeech_lstm_desc.precision of the stacked-LSTM products (recurrent AND batched): exact f32 MFMA (what the reference computes,
# the default and the headline), bf16 hi/lo pairs (three MFMAs per product, ~16 significant bits), plain bf16 (one MFMA, 8 bits:
# BASELINE configs[4]'s "bf16 MFMA").  Gates, cell state, gradients, accumulation, master weights and Adam are f32 in every mode.
PRECISIONS = {"f32": 0, "bf16x3": 1, "bf16": 2}


class Engine(object):
    def __init__(self, num_layers, hidden, input_dim, num_labels, batch_size, max_T, max_U,
                 device="cuda", seed=1234, normalization=False, precision="f32", bidirectional=False,
                 sync_batch_norm=False):
        if not torch.cuda.is_available():
            raise RuntimeError("rnn_speech_amd needs a ROCm GPU (MI355X); there is no CPU path")
        self.L, self.H, self.D, self.C = num_layers, hidden, input_dim, num_labels
        self.B, self.T, self.U = batch_size, max_T, max_U
        self.device = torch.device(device)
        # bidirectional (BASELINE configs[4]; no reference counterpart -- the reference builds a unidirectional dynamic_rnn,
        # :276-278): a second stack of the same shape reads every utterance reversed in time (tf.reverse_sequence semantics,
        # as tf.nn.bidirectional_dynamic_rnn does), the two top outputs are concatenated in front of the output layer
        self.bidirectional = bool(bidirectional)
        self.layout = ParamLayout(num_layers, hidden, input_dim, num_labels, bidirectional=self.bidirectional)
        n = self.layout.total
        self.params = torch.zeros(n, device=self.device)
        self.grads = torch.zeros(n, device=self.device)
        self.adam_m = torch.zeros(n, device=self.device)
        self.adam_v = torch.zeros(n, device=self.device)
        self.norm = torch.zeros(1, device=self.device)
        self.adam_step = 0
        if precision not in PRECISIONS:
            raise ValueError("precision must be 'f32' (exact, default), 'bf16x3' (split-precision MFMA) or 'bf16' (plain bf16 "
                             "operands, f32 accumulation and master weights)")
        self.precision = precision
        self.lstm_ws = ops.LstmWorkspace(max_T, batch_size, hidden, num_layers, device=self.device, precision=PRECISIONS[precision])
        if self.bidirectional:
            self.lstm_ws_b = ops.LstmWorkspace(max_T, batch_size, hidden, num_layers, device=self.device,
                                               precision=PRECISIONS[precision])
            self.ytop_b = torch.empty(max_T, batch_size, hidden, device=self.device)      # backward stack's output, in forward time
            self.dytop_b = torch.empty(max_T, batch_size, hidden, device=self.device)
        self.ctc_ws = ops.CtcWorkspace(max_T, batch_size, num_labels, max_U, self.device)
        self.logits = torch.empty(max_T, batch_size, num_labels, device=self.device)
        self.dlogits = torch.empty_like(self.logits)
        self.loss = torch.zeros(batch_size, device=self.device)
        # persistent RNN state Variables of the reference (:266-275)
        self.state_h = torch.zeros(num_layers, batch_size, hidden, device=self.device)
        self.state_c = torch.zeros(num_layers, batch_size, hidden, device=self.device)
        # optional batch norm of the input-layer output (reference :253-259, off by default)
        self.normalization = bool(normalization)
        # Under data parallelism the moments are taken over THIS rank's mini-batch by default: N ranks x batch b is the
        # reference's mini_batch_size = N accumulation, and the reference normalises every mini-batch with its own moments
        # (:253-259 inside the per-mini-batch graph).  sync_batch_norm=True (opt-in, a DEVIATION from the reference: a
        # different model, three blocking all-reduces per mini-batch, no early stop at the longest utterance) spans the ranks.
        self.sync_batch_norm = bool(sync_batch_norm) and self.normalization
        if self.normalization:
            self.bn_xhat = torch.empty(max_T, batch_size, hidden, device=self.device)
            self.bn_inv_std = torch.empty(max_T, hidden, device=self.device)
            self.bn_scratch = torch.empty(2, max_T, hidden, device=self.device)       # cross-rank sums under data parallelism
        self._ws, self._Tr = self.lstm_ws, max_T
        self._head = None                # ops.CtcHead of the mini-batch in flight, when its CTC stage runs inside the LSTM launches
        self._paired = False             # the last forward ran a bidirectional model's two stacks side by side (ops.lstm_fwd_pair)
        self._ws_b = self.lstm_ws_b if self.bidirectional else None
        # a real (non-NULL) stream for callers that want the overlapped backward pass: see on_stream()
        self.stream = torch.cuda.Stream(device=self.device, priority=-1)      # (ahead of the side stream that prefetches the next batch)
        self._aux_stream = None          # (mini_batch: carries the "beside the forward kernel" ordering point to the caller's hook)
        self._tail_stream = None         # (backward: the dense layers' weight gradients beside the LSTM's)
        self.init_parameters(seed)

    # ---- parameters ------------------------------------------------------------
    def p(self, name):
        return self.layout.view(self.params, name)

    def g(self, name):
        return self.layout.view(self.grads, name)

    def init_parameters(self, seed=1234):
        """Xavier/Glorot-uniform matrices, zero biases (TF defaults, :241-244,302-305)."""
        gen = torch.Generator(device="cpu")
        gen.manual_seed(seed)
        self.params.zero_()
        for name in self.layout.names():
            _, shape = self.layout.slots[name]
            if len(shape) == 2:
                lim = math.sqrt(6.0 / (shape[0] + shape[1]))
                w = (torch.rand(shape, generator=gen) * 2.0 - 1.0) * lim
                self.p(name).copy_(w)

    def load_numpy(self, arrays, flat=None):
        """Copy host arrays into the flat buffer (default: the parameters).  Shapes must match the layout exactly:
        copy_ would silently broadcast e.g. a shape-(1,) bias from a checkpoint of another model size."""
        flat = self.params if flat is None else flat
        for name, a in arrays.items():
            if name not in self.layout.slots:
                raise KeyError("unknown parameter tensor %r (layout has %s)" % (name, ", ".join(self.layout.names())))
            want = tuple(self.layout.slots[name][1])
            got = tuple(np.shape(a))
            if got != want:
                raise ValueError("checkpoint tensor %s has shape %s, the model (L=%d, H=%d, D=%d, C=%d) needs %s"
                                 % (name, got, self.L, self.H, self.D, self.C, want))
            self.layout.view(flat, name).copy_(torch.as_tensor(np.asarray(a), dtype=torch.float32))

    def to_numpy(self, flat=None):
        flat = self.params if flat is None else flat
        return {n: self.layout.view(flat, n).detach().cpu().numpy().copy() for n in self.layout.names()}

    @staticmethod
    def _dp_group():
        from . import dataparallel
        return dataparallel.current()

    # ---- one mini-batch ----------------------------------------------------------
    def _run_length(self, max_len):
        """Frames the recurrence has to visit: the longest utterance of this batch when the host knows it
        (`max_len`, from the dataset pipeline -- no device sync), else the padded length.  Mirrors
        tf.nn.dynamic_rnn, whose while-loop runs to max(sequence_length) (reference :276-278)."""
        if max_len is None:
            return self.T
        if self.sync_batch_norm and self._dp_group().world > 1:
            return self.T          # the batch moments span the ranks: every rank contributes all T frames (padding included)
        return max(1, min(int(max_len), self.T))

    def forward(self, x, lengths, keep_in=1.0, keep_out=1.0, seed=0, use_state=False, max_len=None, after_lstm=None, training=False,
                per_diagonal=False, dense_labels=None):
        """x [T,B,D] device float32, lengths int32 [B] device.  Returns logits [T,B,C]
        (a view of the engine's buffer).  Rows past `max_len` are the output bias (what the
        reference produces there, since the LSTM output is zero past the length)."""
        T, B, D = x.shape
        assert (T, B, D) == (self.T, self.B, self.D), ((T, B, D), (self.T, self.B, self.D))
        x = x.contiguous()
        Tr = self._run_length(max_len)
        ws = self.lstm_ws.prefix(Tr)
        self._ws, self._Tr = ws, Tr
        ws.set_dropout(keep_in, keep_out, seed)
        ops.linear_fwd(x[:Tr].view(Tr * B, D), self.p("input_w"), self.p("input_b"), out=ws.z0.view(Tr * B, self.H))
        if self.normalization:
            grp = self._dp_group() if self.sync_batch_norm else None
            if grp is not None and grp.world > 1:      # moments over the GLOBAL batch: local sums -> all-reduce -> finish (Tr == T on every rank)
                ops.batchnorm_fwd_dp(ws.z0, ws.z0, self.bn_xhat, self.bn_inv_std, grp, self.bn_scratch, 1e-3)
            else:
                ops.batchnorm_fwd(ws.z0, ws.z0, self.bn_xhat[:Tr], self.bn_inv_std[:Tr], 1e-3)
        if self.bidirectional:
            # the backward-direction stack reads the time-reversed input-layer output.  Taken BEFORE the forward stack runs:
            # lstm_fwd applies its layer-0 input-dropout mask to Z_0 in place, and the two stacks' DropoutWrappers are
            # independent (each masks its own copy of the same Z_0)
            wb = self.lstm_ws_b.prefix(Tr)
            self._ws_b = wb
            ops.reverse_sequences(ws.z0, lengths, out=wb.z0)
        # the CTC head inside the LSTM launches (ops.CtcHead; dense_labels given = a training mini-batch): the output layer, the
        # log-softmax and alpha follow the forward recurrence, beta and the gradient lead the backward one
        self._head = None
        if (dense_labels is not None and _FUSED_CTC and not self.bidirectional
                and ops.lstm_ctc_fusable(ws, self.C, dense_labels.shape[1], per_diagonal=per_diagonal)):
            self._head = ops.CtcHead(self.p("output_w"), self.p("output_b"), self.logits[:Tr], dense_labels, self.loss,
                                     self.dlogits[:Tr], self.ctc_ws)
        paired = self.bidirectional and _BIDIR_PAIR and not per_diagonal and ops.lstm_pair_fusable(ws)
        self._paired = bool(paired)
        if paired:
            # the two stacks in ONE call: where the forward kernel places a batch tile's group on one XCD (1024 wide in plain bf16)
            # their layers run side by side, one launch per layer for both (ops.lstm_fwd_pair)
            wb.set_dropout(keep_in, keep_out, seed ^ 0x5bd1e995)
            ops.lstm_fwd_pair(ws, self.p("kernel_0"), self.p("bias_0"), wb, self.p("bw_kernel_0"), self.p("bw_bias_0"),
                              self.layout.kernel_stride, self.layout.bias_stride, lengths,
                              self.state_h if use_state else None, self.state_c if use_state else None)
        else:
            ops.lstm_fwd(ws, self.p("kernel_0"), self.layout.kernel_stride, self.p("bias_0"),
                         self.layout.bias_stride, lengths,
                         self.state_h if use_state else None, self.state_c if use_state else None, training=training,
                         per_diagonal=per_diagonal, head=self._head)
        H = self.H
        if not self.bidirectional:
            if after_lstm is not None:
                after_lstm()           # (mini_batch: from here on other streams may use the chip -- see beside_ctc)
            if self._head is None:
                ops.linear_fwd(ws.ztop.view(Tr * B, H), self.p("output_w"), self.p("output_b"),
                               out=self.logits[:Tr].view(Tr * B, self.C))
        else:
            # backward-direction stack on the time-reversed input-layer output (its own dropout stream; it always starts
            # from a zero state: a state carried from the END of the previous batch's utterances means nothing here)
            if not paired:
                wb.set_dropout(keep_in, keep_out, seed ^ 0x5bd1e995)
                ops.lstm_fwd(wb, self.p("bw_kernel_0"), self.layout.kernel_stride, self.p("bw_bias_0"),
                             self.layout.bias_stride, lengths, None, None, training=training, per_diagonal=per_diagonal)
            if after_lstm is not None:
                after_lstm()
            ops.reverse_sequences(wb.ztop, lengths, out=self.ytop_b[:Tr])
            wo = self.p("output_w")
            ops.linear_fwd(ws.ztop.view(Tr * B, H), wo[:H], self.p("output_b"), out=self.logits[:Tr].view(Tr * B, self.C))
            ops.gemm(self.ytop_b[:Tr].view(Tr * B, H), wo[H:], out=self.logits[:Tr].view(Tr * B, self.C), accumulate=True)
        if Tr < T:
            self.logits[Tr:] = self.p("output_b")          # broadcast fill of the never-visited tail
        return self.logits

    def kernel_path(self):
        """Which shape-driven choices the LAST forward made -- what a parity test has to assert before it may claim to have checked
        them: `fused_ctc_head` (the CTC stage ran inside the two whole-sequence LSTM launches, ops.CtcHead), `paired` (a
        bidirectional model's two stacks side by side on one-XCD groups, ops.lstm_fwd_pair), `run_length` (frames visited)."""
        return {"fused_ctc_head": self._head is not None, "paired": self._paired, "run_length": self._Tr}

    def final_state(self):
        """(h [L,B,H], c [L,B,H]) after the last forward."""
        return self._ws.final_state()

    def keep_state(self):
        """rnn_keep_state_op (:281-289): final state -> persistent state."""
        h, c = self._ws.final_state()
        self.state_h.copy_(h)
        self.state_c.copy_(c)

    def zero_state(self):
        self.state_h.zero_()
        self.state_c.zero_()

    def ctc(self, dense_labels, lengths, stage=0):
        Tr = self._Tr
        if self._head is not None:       # (the loss is the forward launch's; dlogits will be the backward launch's)
            if Tr < self.T and stage != 1:
                self.dlogits[Tr:].zero_()
            return self.loss
        ops.ctc_loss_fwd_bwd(self.logits[:Tr], dense_labels, lengths, ws=self.ctc_ws, loss=self.loss,
                             dlogits=self.dlogits[:Tr], stage=stage)
        if Tr < self.T and stage != 1:
            self.dlogits[Tr:].zero_()
        return self.loss

    @contextlib.contextmanager
    def on_stream(self):
        """Run the enclosed engine calls on the engine's own (non-default) stream, ordered after the
        caller's current stream and joined back into it on exit.  The backward pass overlaps the weight-
        gradient GEMMs with the BPTT chain on CU-partitioned HIP streams; those are "blocking" streams, so
        the overlap is only used (and only pays off) when the work is NOT issued on the legacy NULL stream
        (csrc/lstm.hip, lstm_bwd).  A caller that already sits on a real stream is left there."""
        cur = torch.cuda.current_stream(self.device)
        if cur.cuda_stream != 0:
            yield
            return
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            yield
        cur.wait_stream(self.stream)

    def backward(self, x, lengths, wait_for=None, per_diagonal=False):
        """Accumulates d(sum_b loss_b)/d(theta) into self.grads (for the batch of the last forward).
        wait_for: an event the backward RECURRENCE has to wait for (side-stream work placed beside the CTC stage)."""
        T, B, D = x.shape
        x = x.contiguous()
        ws, Tr = self._ws, self._Tr
        H = self.H
        dl = self.dlogits[:Tr].view(Tr * B, self.C)
        head = self._head if not per_diagonal else None
        if self._head is not None and head is None:
            raise _lib.AmdSpeechError("backward(per_diagonal=True) after a forward with the fused CTC head: repeat the forward too")
        if head is not None:
            pass                         # dlogits and dZ_top are formed inside lstm_bwd; dW_o / db_o follow it
        elif not self.bidirectional:
            ops.linear_bwd(ws.ztop.view(Tr * B, H), self.p("output_w"), dl,
                           self.g("output_w"), self.g("output_b"), need_dx=True, dx=ws.dztop.view(Tr * B, H))
        else:
            wo, gwo, wb = self.p("output_w"), self.g("output_w"), self._ws_b
            ops.linear_bwd(ws.ztop.view(Tr * B, H), wo[:H], dl, gwo[:H], self.g("output_b"), need_dx=True,
                           dx=ws.dztop.view(Tr * B, H))
            yb = self.ytop_b[:Tr].view(Tr * B, H)
            ops.gemm(yb, dl, trans_a=True, out=gwo[H:], accumulate=True)                     # dW_o[H:] += y_b^T . dlogits
            ops.gemm(dl, wo[H:], trans_b=True, out=self.dytop_b[:Tr].view(Tr * B, H))        # d y_b = dlogits . W_o[H:]^T
            ops.reverse_sequences(self.dytop_b[:Tr], lengths, out=wb.dztop)
        if wait_for is not None:
            torch.cuda.current_stream(self.device).wait_event(wait_for)
        paired = self.bidirectional and _BIDIR_PAIR and not per_diagonal and ops.lstm_pair_fusable(ws)
        if paired:                       # the two stacks' layers side by side (ops.lstm_bwd_pair)
            wb = self._ws_b
            ops.lstm_bwd_pair(ws, self.p("kernel_0"), self.g("kernel_0"), self.g("bias_0"),
                              wb, self.p("bw_kernel_0"), self.g("bw_kernel_0"), self.g("bw_bias_0"),
                              self.layout.kernel_stride, self.layout.bias_stride, lengths)
        else:
            ops.lstm_bwd(ws, self.p("kernel_0"), self.layout.kernel_stride, self.g("kernel_0"), self.g("bias_0"),
                         self.layout.bias_stride, lengths, per_diagonal=per_diagonal, head=head)
        # the dense layers' weight gradients need nothing but the backward kernel's results: beside the first of the weight-gradient
        # launches that follow it (ops.lstm_beside_tail) instead of behind the last
        side = None
        if _BESIDE_TAIL and not self.bidirectional and not self.normalization and not per_diagonal:
            if self._tail_stream is None:
                self._tail_stream = torch.cuda.Stream(self.device)
            flags = ops.lstm_beside_tail(ws, self._tail_stream)
            if flags & 1:
                side = self._tail_stream
                with torch.cuda.stream(side):
                    if head is not None:
                        ops.linear_bwd(ws.ztop.view(Tr * B, H), self.p("output_w"), dl, self.g("output_w"), self.g("output_b"), need_dx=False)
                    if flags & 2:
                        ops.linear_bwd(x[:Tr].view(Tr * B, D), self.p("input_w"), ws.dz0.view(Tr * B, self.H), self.g("input_w"),
                                       self.g("input_b"), need_dx=False)
                torch.cuda.current_stream(self.device).wait_stream(side)
                if flags & 2:
                    return
                head = None
        if head is not None:             # dW_o += Z_top^T . dlogits, db_o += column sums of dlogits
            ops.linear_bwd(ws.ztop.view(Tr * B, H), self.p("output_w"), dl, self.g("output_w"), self.g("output_b"), need_dx=False)
        if self.bidirectional:
            wb = self._ws_b
            if not paired:
                ops.lstm_bwd(wb, self.p("bw_kernel_0"), self.layout.kernel_stride, self.g("bw_kernel_0"), self.g("bw_bias_0"),
                             self.layout.bias_stride, lengths, per_diagonal=per_diagonal)
            ops.reverse_sequences(wb.dz0, lengths, out=ws.dz0, accumulate=True)              # both stacks read the same Z_0
        if self.normalization:
            grp = self._dp_group() if self.sync_batch_norm else None
            if grp is not None and grp.world > 1:
                ops.batchnorm_bwd_dp(ws.dz0, self.bn_xhat, self.bn_inv_std, ws.dz0, grp, self.bn_scratch)
            else:
                ops.batchnorm_bwd(ws.dz0, self.bn_xhat[:Tr], self.bn_inv_std[:Tr], ws.dz0)
        ops.linear_bwd(x[:Tr].view(Tr * B, D), self.p("input_w"), ws.dz0.view(Tr * B, self.H), self.g("input_w"),
                       self.g("input_b"), need_dx=False)

    def check(self):
        """Synchronous health check of the dataflow LSTM kernels: their waits are bounded, and a time-out (the
        workgroups of one launch were not all resident -- another kernel held CUs) leaves an error flag behind
        instead of hanging.  Raises AmdSpeechError; results of that step are invalid."""
        ops.lstm_status(self._ws)
        if self.bidirectional:
            ops.lstm_status(self._ws_b)

    def healthy(self):
        """check() as a predicate: False when a dataflow launch of the last mini-batch gave up waiting (its results -- logits, loss,
        the gradient contribution, the final state -- are invalid; see mini_batch(per_diagonal=True) for the way out).  Only the
        time-out is recoverable: any other error of the status call (a sticky HIP fault) is raised, not turned into a retry."""
        try:
            self.check()
            return True
        except _lib.DataflowTimeout:
            return False

    def zero_grads(self):
        self.grads.zero_()

    def mini_batch(self, x, lengths, dense_labels, keep_in=1.0, keep_out=1.0, seed=0, use_state=False,
                   compute_gradients=True, max_len=None, beside_ctc=None, marks=None, beside_forward=None, per_diagonal=False):
        """forward -> CTC -> backward, with two slots for side work on other streams; each is an optional
        callable(after_event) -> done_event (or None) that enqueues short-lived kernels / copies ordered after `after_event`:
          beside_forward: work that needs NOTHING of this mini-batch (the next one's front end).  When the forward recurrence
            is the whole-sequence dataflow kernel with XCDs to spare, `after_event` is the point just in front of its launch
            (ops.lstm_beside_forward): the front end's work-queue kernel then does its work on the idle XCDs while the
            recurrence runs (and completes with it).  Otherwise the call comes in the other slot;
          beside_ctc: work that needs this mini-batch's LOGITS (the training-time decoder's copy): beside the CTC recursions
            between the two recurrence kernels (64 of the 256 CUs busy for ~0.2 ms).
        The backward recurrence waits for both.  Nothing is ever placed beside the backward kernel: it keeps every CU.
        per_diagonal: run the stack on the launch-per-diagonal kernels (amdspeech.h: AMDSPEECH_LSTM_PER_DIAGONAL) -- the repeat of
        a mini-batch whose whole-sequence launch timed out (healthy() is False); same results, no co-residency requirement."""
        def mark(name):                # (timeline: HIP events between the stages, read by the caller after a sync)
            if marks is not None:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record(torch.cuda.current_stream(self.device))
                marks.append((name, ev))

        mark("begin")
        placed = []

        def try_beside_forward():
            # the XCDs the whole-sequence forward kernel leaves without a recurrence group (two of eight for ~5 ms at 3x512 /
            # batch 32): amdspeech_lstm_beside_forward.  Beside the CTC recursions the same work shares SIMDs with them
            # (alpha/beta 150 -> 250 us with the matrix-core front end next to it).
            if beside_forward is None or not _BESIDE_FORWARD:
                return
            if self._aux_stream is None:
                self._aux_stream = torch.cuda.Stream(self.device)
            if ops.lstm_beside_forward(self._ws, self._aux_stream) > 0:
                after = torch.cuda.Event()
                after.record(self._aux_stream)
                placed.append(beside_forward(after))

        self.forward(x, lengths, keep_in, keep_out, seed, use_state, max_len, training=compute_gradients, after_lstm=try_beside_forward,
                     per_diagonal=per_diagonal, dense_labels=dense_labels)
        mark("forward")
        pending = [ev for ev in placed if ev is not None]
        late = [h for h in (beside_ctc, (beside_forward if not placed else None)) if h is not None]
        if self._head is not None:
            # The CTC stage ran inside the forward launch and will finish inside the backward one: there is no stage to place work
            # beside.  Hooks that need the logits are ordered behind the forward launch and JOINED BEHIND the backward pass: short
            # kernels and copies that become runnable together with the backward kernel either slip in front of it (the library's
            # own side-stream fills keep that launch waiting ~0.1 ms anyway) or run when it ends -- they depend on nothing of it,
            # so its workgroups never wait for more than their duration.
            after = torch.cuda.Event()
            after.record(torch.cuda.current_stream(self.device))
            for h in late:
                ev = h(after)
                if ev is not None:
                    pending.append(ev)
            self.ctc(dense_labels, lengths)
            mark("ctc")
            if compute_gradients:
                self.backward(x, lengths, per_diagonal=per_diagonal)
                mark("backward")
            cur = torch.cuda.current_stream(self.device)
            for ev in pending:
                cur.wait_event(ev)
            return self.loss
        if late:
            # behind the output layer and the log-softmax (both fill the chip and are short), beside the CTC recursions
            self.ctc(dense_labels, lengths, stage=1)
            after = torch.cuda.Event()
            after.record(torch.cuda.current_stream(self.device))
            for h in late:
                ev = h(after)
                if ev is not None:
                    pending.append(ev)
            self.ctc(dense_labels, lengths, stage=2)
        else:
            self.ctc(dense_labels, lengths)
        mark("ctc")
        cur = torch.cuda.current_stream(self.device)
        for ev in pending[:-1]:
            cur.wait_event(ev)
        done = pending[-1] if pending else None
        if compute_gradients:
            self.backward(x, lengths, wait_for=done, per_diagonal=per_diagonal)
            mark("backward")
        elif done is not None:
            cur.wait_event(done)      # the next recurrence kernel must not start beside it
        return self.loss

    # ---- optimiser step ------------------------------------------------------------
    def all_reduce_grads(self):
        """Data parallel: ONE fused fp32 SUM all-reduce of the flat gradient buffer (amdspeech_allreduce_sum_f32 =
        RCCL over xGMI, see dataparallel.py); N ranks x batch b is then the reference's mini_batch_size=N
        accumulation (:391-406).  A no-op in a single-process job."""
        from . import dataparallel
        dataparallel.current().all_reduce_sum_(self.grads)

    def broadcast_state(self, root=0):
        """Make every replica bit-identical to rank `root`: parameters and Adam moments (after a restore)."""
        from . import dataparallel
        grp = dataparallel.current()
        for flat in (self.params, self.adam_m, self.adam_v):
            grp.broadcast_(flat, root)
        self.adam_step = int(grp.broadcast_object(self.adam_step, root))

    def apply(self, lr, clip, beta1=0.9, beta2=0.999, eps=1e-8):
        self.adam_step += 1
        t = self.adam_step
        lr_t = lr * math.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
        ops.clip_adam(self.params, self.grads, self.adam_m, self.adam_v, float(clip), lr_t, beta1, beta2, eps,
                      norm_out=self.norm)
        return self.norm
