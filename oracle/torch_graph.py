"""TEST / BASELINE INFRASTRUCTURE ONLY -- never imported by the product (rnn-speech_amd/).

A torch-CPU restatement of the reference's training graph, used (a) by bench.py's `cpu_baseline`
leg as the closest available analogue of the reference's TensorFlow-CPU (Eigen/MKL) kernels -- SURVEY.md
section 8(d) "(ii) a torch-CPU restatement of the identical graph" -- and (b) by tests/ as a second,
independent checker of the numpy oracle (oracle/model.py).  TensorFlow itself cannot be installed here
(no network, TF-1.x API), so like oracle/model.py this is a PORT: **parity unpinned** against the real
TensorFlow ops; it is pinned only against the numpy oracle (tests/test_cpu_oracle.py).

Graph restated (reference models/AcousticModel.py):
  input layer      x[t].W_i + b_i                                   :240-250
  stacked LSTM     BasicLSTMCell (gates i,j,f,o; forget_bias 1), MultiRNNCell, dynamic_rnn with
                   sequence_length (zero output / state copy past the length)      :223-237, :276-278
  output layer     y[t].W_o + b_o                                   :301-309
  CTC              tf.nn.ctc_loss, summed over the batch by compute_gradients      :356-361, :386-388
  optimiser        clip_by_global_norm + Adam (TF form)                            :388, :404-406
torch.nn.LSTM keeps its gates in the order (i, f, g, o) and carries two bias vectors; the TensorFlow
kernel K [2H,4H] (rows x then h, column blocks i|j|f|o) is split and permuted accordingly, and the forget
bias goes into b_ih.  Packed sequences give dynamic_rnn's length semantics.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


def _perm(H):
    # torch blocks (i, f, g, o)  <-  TensorFlow blocks (i, j, f, o)
    idx = np.arange(4 * H).reshape(4, H)
    return torch.as_tensor(np.concatenate([idx[0], idx[2], idx[1], idx[3]]))


class TorchGraph(object):
    """Parameters are taken / returned in the oracle's dict layout (input_w, input_b, kernel_l, bias_l,
    output_w, output_b) so the two checkers can be compared directly."""

    def __init__(self, params, num_layers, dtype=torch.float32):
        self.L = num_layers
        self.dtype = dtype
        self.p = {k: torch.tensor(np.asarray(v), dtype=dtype, requires_grad=True) for k, v in params.items()}
        self.H = self.p["input_w"].shape[1]
        self.C = self.p["output_w"].shape[1]
        self.m = {k: torch.zeros_like(v) for k, v in self.p.items()}
        self.v = {k: torch.zeros_like(v) for k, v in self.p.items()}
        self.step = 0
        self._perm = _perm(self.H)

    def _lstm_weights(self, l):
        H, perm = self.H, self._perm
        K, b = self.p["kernel_%d" % l], self.p["bias_%d" % l]
        w_ih = K[:H][:, perm].t().contiguous()
        w_hh = K[H:][:, perm].t().contiguous()
        fb = torch.zeros(4 * H, dtype=self.dtype)
        fb[H:2 * H] = 1.0                                    # forget_bias (torch block 1 = f)
        return w_ih, w_hh, b[perm] + fb, torch.zeros(4 * H, dtype=self.dtype)

    def forward(self, x, lengths):
        """x [T,B,D] float array, lengths [B] ints -> logits [T,B,C] (torch, with graph)."""
        x = torch.as_tensor(np.asarray(x), dtype=self.dtype)
        T, B, _ = x.shape
        lens = torch.as_tensor(np.minimum(np.asarray(lengths), T).astype(np.int64))
        y = x @ self.p["input_w"] + self.p["input_b"]
        zeros = lambda n: torch.zeros(1, n, self.H, dtype=self.dtype)      # noqa: E731
        if bool((lens == T).all()):
            # every utterance spans the padded length: plain (un-packed) sequences, the fast oneDNN/MKL path
            out = y
            for l in range(self.L):
                out = torch._VF.lstm(out, (zeros(B), zeros(B)), list(self._lstm_weights(l)), True, 1, 0.0,
                                     False, False, False)[0]
            return out @ self.p["output_w"] + self.p["output_b"]
        keep = lens > 0
        out = torch.zeros(T, B, self.H, dtype=self.dtype)
        if bool(keep.any()):
            idx = torch.nonzero(keep).flatten()
            seq = y[:, idx]
            for l in range(self.L):
                packed = torch.nn.utils.rnn.pack_padded_sequence(seq, lens[idx], enforce_sorted=False)
                flat = torch._VF.lstm(packed.data, packed.batch_sizes, (zeros(len(idx)), zeros(len(idx))),
                                      list(self._lstm_weights(l)), True, 1, 0.0, False, False)
                packed = torch.nn.utils.rnn.PackedSequence(flat[0], packed.batch_sizes, packed.sorted_indices,
                                                           packed.unsorted_indices)
                seq, _ = torch.nn.utils.rnn.pad_packed_sequence(packed, total_length=T)
            out = out.index_copy(1, idx, seq)
        return out @ self.p["output_w"] + self.p["output_b"]

    def loss(self, logits, label_rows, lengths):
        """Per-utterance CTC loss [B]; label_rows = oracle.model.sparsify_labels(...).  TensorFlow's
        conventions (oracle.model.ctc_targets): blank = C-1, the target ends at the first label >= C-1
        (the EOS token), rows whose RAW label count exceeds the frame count are ignored."""
        T, B, C = logits.shape
        lens = np.minimum(np.asarray(lengths), T).astype(np.int64)
        targets, valid = [], []
        for b, r in enumerate(label_rows):
            tgt = []
            for v in r:
                if v >= C - 1:
                    break
                tgt.append(int(v))
            targets.append(np.asarray(tgt, np.int64))
            valid.append(lens[b] > 0 and len(r) <= lens[b])
        valid = torch.as_tensor(np.asarray(valid))
        tl = torch.as_tensor([len(t) for t in targets])
        cat = torch.as_tensor(np.concatenate(targets) if targets else np.zeros(0, np.int64))
        logp = F.log_softmax(logits, dim=2)
        per = F.ctc_loss(logp, cat, torch.as_tensor(np.maximum(lens, 1)), tl, blank=C - 1, reduction="none",
                         zero_infinity=False)
        return torch.where(valid, per, torch.zeros_like(per))

    def train_step(self, x, lengths, label_rows, lr, clip, beta1=0.9, beta2=0.999, eps=1e-8):
        for v in self.p.values():
            v.grad = None
        logits = self.forward(x, lengths)
        per = self.loss(logits, label_rows, lengths)
        per.sum().backward()
        with torch.no_grad():
            g = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in self.p.items()}
            norm = math.sqrt(sum(float((a.double() ** 2).sum()) for a in g.values()))
            scale = clip / max(norm, clip)
            self.step += 1
            lr_t = lr * math.sqrt(1.0 - beta2 ** self.step) / (1.0 - beta1 ** self.step)
            for k, v in self.p.items():
                gk = g[k] * scale
                self.m[k].mul_(beta1).add_(gk, alpha=1.0 - beta1)
                self.v[k].mul_(beta2).addcmul_(gk, gk, value=1.0 - beta2)
                v.sub_(lr_t * self.m[k] / (self.v[k].sqrt() + eps))
        return per.detach().numpy(), norm, logits.detach().numpy()

    def grads(self):
        return {k: v.grad.detach().numpy().copy() for k, v in self.p.items()}

    def params(self):
        return {k: v.detach().numpy().copy() for k, v in self.p.items()}
