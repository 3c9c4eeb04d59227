"""TEST INFRASTRUCTURE: an Engine whose numerics are the numpy oracle, on CPU tensors.

Lets the multi-process (gloo, no GPU) tests drive the PRODUCT's host logic -- AcousticModel.run_train_step /
end_batch / save / restore, Engine.all_reduce_grads / broadcast_state, dataparallel.Group -- end to end.  The kernels
themselves are covered by the `-m gpu` tests; nothing here is importable from the product path."""
import contextlib

import numpy as np
import torch

from oracle import model as om
from rnn_speech_amd.acoustic_model import AcousticModel
from rnn_speech_amd.engine import Engine, ParamLayout


class OracleEngine(Engine):
    def __init__(self, num_layers, hidden, input_dim, num_labels, batch_size, max_T, max_U, seed=1234, **_):
        self.L, self.H, self.D, self.C = num_layers, hidden, input_dim, num_labels
        self.B, self.T, self.U = batch_size, max_T, max_U
        self.device = torch.device("cpu")
        self.layout = ParamLayout(num_layers, hidden, input_dim, num_labels)
        n = self.layout.total
        self.params, self.grads = torch.zeros(n), torch.zeros(n)
        self.adam_m, self.adam_v = torch.zeros(n), torch.zeros(n)
        self.norm = torch.zeros(1)
        self.adam_step = 0
        self.logits = torch.zeros(max_T, batch_size, num_labels)
        self.loss = torch.zeros(batch_size)
        self.state_h = torch.zeros(num_layers, batch_size, hidden)
        self.state_c = torch.zeros(num_layers, batch_size, hidden)
        self.normalization = False
        self.init_parameters(seed)

    @contextlib.contextmanager
    def on_stream(self):
        yield

    def _p64(self):
        return {k: v.astype(np.float64) for k, v in self.to_numpy().items()}

    def mini_batch(self, x, lengths, dense_labels, keep_in=1.0, keep_out=1.0, seed=0, use_state=False,
                   compute_gradients=True, max_len=None, beside_ctc=None, marks=None, beside_forward=None, per_diagonal=False):
        self.per_diagonal_runs = getattr(self, "per_diagonal_runs", 0) + (1 if per_diagonal else 0)
        for hook in (beside_ctc, beside_forward):
            if hook is not None:
                hook(None)                     # the product's side-work hooks (host half only on CPU)
        x = np.asarray(x, np.float64)
        lengths = np.asarray(lengths)
        dense = np.asarray(dense_labels)
        p = self._p64()
        logits, _, cache = om.forward(p, x, lengths, self.L, keep_cache=True)      # (dropout / state carry: not modelled)
        loss, dl = om.ctc_loss_and_grad(logits, om.sparsify_labels(dense, self.C), lengths)
        self.logits.copy_(torch.as_tensor(logits, dtype=torch.float32))
        self.loss.copy_(torch.as_tensor(loss, dtype=torch.float32))
        if compute_gradients:
            g = om.backward(p, cache, dl, lengths, self.L)
            for k, v in g.items():
                self.layout.view(self.grads, k).add_(torch.as_tensor(v, dtype=torch.float32))
        return self.loss

    def forward(self, x, lengths, *a, **kw):
        logits, _, _ = om.forward(self._p64(), np.asarray(x, np.float64), np.asarray(lengths), self.L)
        self.logits.copy_(torch.as_tensor(logits, dtype=torch.float32))
        return self.logits

    def keep_state(self):
        pass

    def check(self):
        # tests: the next `fail_checks` health checks report a dataflow time-out (what amdspeech_lstm_status does on a GPU)
        if getattr(self, "fail_checks", 0) > 0:
            self.fail_checks -= 1
            from rnn_speech_amd.lib import DataflowTimeout
            raise DataflowTimeout("injected time-out")

    def apply(self, lr, clip, beta1=0.9, beta2=0.999, eps=1e-8):
        self.adam_step += 1
        names = self.layout.names()
        p = {k: self.layout.view(self.params, k).numpy() for k in names}          # views: updated in place
        g = {k: self.layout.view(self.grads, k).numpy() for k in names}
        m = {k: self.layout.view(self.adam_m, k).numpy() for k in names}
        v = {k: self.layout.view(self.adam_v, k).numpy() for k in names}
        gn = om.clip_and_adam(p, g, m, v, self.adam_step, lr, clip, beta1, beta2, eps)
        for k in names:                        # clip_and_adam rebinds m[k] / v[k]
            self.layout.view(self.adam_m, k).copy_(torch.as_tensor(m[k]))
            self.layout.view(self.adam_v, k).copy_(torch.as_tensor(v[k]))
        self.norm[0] = gn
        return self.norm


class OracleAcousticModel(AcousticModel):
    def _make_engine(self):
        self.engine = OracleEngine(self.num_layers, self.hidden_size, self.input_dim, self.num_labels,
                                   self.batch_size, self.max_input_seq_length, self.max_target_seq_length)
        self.rnn_created = True

    def _error_rate_launch(self, dlen, dense):      # -> (distances [B] with a .cpu().numpy(), truth lengths), as the product's
        lengths = np.asarray(dlen)
        ids = om.greedy_decode(self.engine.logits.numpy(), lengths)
        rows = om.sparsify_labels(dense, self.num_labels)
        import torch
        dist = torch.as_tensor([float(om.edit_distance(i, r)) for i, r in zip(ids, rows)])
        return dist, np.asarray([len(r) for r in rows], np.float64)


class ListDataset(object):
    """A dataset of ready-made mini-batches [(feat [T,B,D] float32, lengths int32 [B], dense int32 [B,U]), ...]."""

    def __init__(self, batches):
        self._batches = list(batches)

    def batches(self):
        for b in self._batches:
            yield b

    def with_items(self, batches):
        return ListDataset(batches)
