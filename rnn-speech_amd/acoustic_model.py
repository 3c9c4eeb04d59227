"""AcousticModel drop-in: the class surface `stt.py` drives in the reference
(/root/reference/models/AcousticModel.py, signatures in SURVEY.md Appendix B), with the
TensorFlow graph replaced by rnn_speech_amd.engine.Engine (HIP kernels behind the C ABI).

`session` / `run_options` / `run_metadata` arguments are accepted and ignored; a tiny
`Session` shim runs the two "ops" stt.py executes directly (`learning_rate_decay_op`,
iterator initialisers).  End of dataset is reported through `dataset_empty=True`, as the
reference does after catching tf.errors.OutOfRangeError (:921-923).
"""
import logging
import os
import time
from random import randint

import collections

import numpy as np
import torch

from . import dataparallel
from . import labels as _labels
from . import ops
from .audioprocessor import AudioProcessor
from .engine import Engine


class OutOfRangeError(Exception):
    """Stands in for tf.errors.OutOfRangeError (iterator exhausted)."""


class _Op(object):
    def __init__(self, fn):
        self.fn = fn

    def __call__(self):
        return self.fn()


class Session(object):
    """Minimal stand-in for tf.Session: `run(op)` calls the op (or each op of a list)."""

    def run(self, op, *args, **kwargs):
        if isinstance(op, (list, tuple)):
            return [self.run(o) for o in op]
        return op() if callable(op) else op

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


class _Variable(object):
    def __init__(self, value):
        self.value = value

    def eval(self, session=None):
        return self.value


# --------------------------------------------------------------------- datasets
class AcousticDataset(object):
    """What build_dataset returns: items [audio, label, (length)] where `audio` is a file
    path or an in-memory (signal, sample_rate) pair, batched to [T_max, B, D] device tensors
    with the reference's padding rules (:825-827, :144-159).

    The reference maps files through two py_func threads and prefetches 30 items (:820-822), and recomputes
    every feature every epoch.  Here a producer thread decodes the NEXT `prefetch` mini-batches (native
    decoders on a thread pool, GIL released) while the GPU trains on the current one, and -- optionally --
    the features of file items are kept in host memory after their first use (`feature_cache_mb` > 0), so
    later epochs skip decode, resampling and the front end altogether."""

    def __init__(self, input_set, batch_size, max_input_seq_length, max_target_seq_length,
                 signal_processing, char_map, n_mfcc=20, device="cuda", prefetch=2, feature_cache_mb=0,
                 sample_rate=22050):
        self.items = [(it[0], it[1]) for it in input_set]
        self.batch_size = batch_size
        self.T = max_input_seq_length
        self.U = max_target_seq_length
        self.char_map = char_map
        self.audio = AudioProcessor(max_input_seq_length, signal_processing, n_mfcc=n_mfcc, device=device,
                                    load_sr=sample_rate)
        self.prefetch = int(prefetch)
        self._signal_processing, self._n_mfcc = signal_processing, n_mfcc
        self._cache = {} if feature_cache_mb > 0 else None
        self._room = [int(feature_cache_mb) << 20]          # shared (by reference) with reordered siblings

    def with_items(self, input_set):
        """The same dataset over a re-ordered / re-shuffled item list, sharing the feature cache (the
        reference builds a fresh tf.data pipeline at every epoch, stt.py:198-207)."""
        other = AcousticDataset(input_set, self.batch_size, self.T, self.U, self._signal_processing, self.char_map,
                                n_mfcc=self._n_mfcc, device=self.audio.device, prefetch=self.prefetch,
                                sample_rate=self.audio.load_sr)
        other._cache, other._room = self._cache, self._room
        return other

    # ---- producer side (host only) -------------------------------------------------
    def _chunks(self):
        B = self.batch_size
        for start in range(0, len(self.items), B):
            yield self.items[start:start + B]

    def _prepare(self, chunk):
        """Host half of one mini-batch: cached features or decoded waveforms per item."""
        from .audioprocessor import decode_files
        cache = self._cache
        need = [a for a, _ in chunk if isinstance(a, str) and (cache is None or a not in cache)]
        from_disk = dict(zip(need, decode_files(need)))
        parts = []
        for a, _ in chunk:
            if not isinstance(a, str):
                parts.append(("signal", (np.asarray(a[0], np.float32), int(a[1]))))
            elif a in from_disk:
                parts.append(("decoded", from_disk[a]))
            else:
                parts.append(("cached", cache[a]))
        # everything the consumer would otherwise do on the training thread between two kernel launches: the label codec
        # (0.12 ms per utterance) and the packing of the waveforms into one pinned block for an asynchronous upload
        B = self.batch_size
        dense = np.zeros((B, self.U), np.int32)
        for i, (_, text) in enumerate(chunk):
            ids = _labels.get_str_labels(self.char_map, text)[:self.U]
            dense[i, :len(ids)] = ids
        kinds = {k for k, _ in parts}
        staged = None
        if kinds == {"signal"} and len({p[1] for _, p in parts}) == 1:
            sigs = [p[0] for _, p in parts] + [np.zeros(0, np.float32)] * (B - len(parts))
            staged = ("batch", self.audio.stage(sigs), parts[0][1][1])
        elif "cached" not in kinds:
            staged = ("files", self.audio.stage_files([p for _, p in parts]))
        return chunk, parts, dense, staged

    def _prepared(self):
        if self.prefetch <= 0:
            for chunk in self._chunks():
                yield self._prepare(chunk)
            return
        import queue
        import threading
        q = queue.Queue(maxsize=self.prefetch)
        stop = threading.Event()

        def produce():
            try:
                for chunk in self._chunks():
                    if stop.is_set():
                        return
                    q.put(("ok", self._prepare(chunk)))
                q.put(("end", None))
            except BaseException as exc:            # surfaced in the consumer
                q.put(("error", exc))

        th = threading.Thread(target=produce, name="amdspeech-decode", daemon=True)
        th.start()
        try:
            while True:
                tag, payload = q.get()
                if tag == "end":
                    return
                if tag == "error":
                    raise payload
                yield payload
        finally:
            stop.set()
            while th.is_alive() or not q.empty():   # unblock a producer waiting on a full queue; hand its staging blocks back
                try:
                    tag, payload = q.get_nowait()
                    if tag == "ok":
                        self._release(payload[3])
                except queue.Empty:
                    th.join(0.01)

    @staticmethod
    def _release(staged):
        """A prepared mini-batch that will never be uploaded (the iterator was reset): its pinned blocks return to the pool."""
        if staged is None:
            return
        blocks = [staged[1][0]] if staged[0] == "batch" else [part[2] for part in staged[1]]
        for blk in blocks:
            blk.claimed = False

    # ---- consumer side (device) -----------------------------------------------------
    def batches(self):
        B, T = self.batch_size, self.T
        for chunk, parts, dense, staged in self._prepared():
            if staged is not None and staged[0] == "batch":
                # in-memory signals at one rate: the process_signal convention (no resampling)
                feat, lengths = self.audio.process_batch(None, staged[2], t_max=T, staged=staged[1])
            elif staged is not None:
                # files: librosa.load semantics (22,050 Hz); a short final batch is padded with empty rows
                feat, lengths = self.audio.process_files(None, t_max=T, rows=B, decoded=[p for _, p in parts], staged=staged[1])
            else:
                feat, lengths = self._assemble(parts)
            if self._cache is not None:
                self._remember(chunk, parts, feat, lengths)
            yield feat, np.asarray(lengths, np.int32), dense

    def _assemble(self, parts):
        """A mini-batch with cached rows: fresh rows go through the device path, cached rows are copied in."""
        B, T, D = self.batch_size, self.T, self.audio.feature_size
        fresh = [i for i, (k, _) in enumerate(parts) if k != "cached"]
        lengths = [0] * B
        host = np.zeros((T, B, D), np.float32)
        for i, (k, p) in enumerate(parts):
            if k == "cached":
                f, n = p
                host[:f.shape[0], i] = f
                lengths[i] = n
        feat = torch.from_numpy(host).to(self.audio.device)
        if fresh:
            sub, sub_len = self.audio.process_files(None, t_max=T, decoded=[parts[i][1] for i in fresh])
            feat[:, torch.as_tensor(fresh, device=feat.device)] = sub
            for j, i in enumerate(fresh):
                lengths[i] = sub_len[j]
        return feat, lengths

    def _remember(self, chunk, parts, feat, lengths):
        rows = [i for i, (k, _) in enumerate(parts) if k == "decoded"]
        if not rows or self._room[0] <= 0:
            return
        host = feat[:, torch.as_tensor(rows, device=feat.device)].cpu().numpy()
        for j, i in enumerate(rows):
            n = min(int(lengths[i]), self.T)
            f = host[:n, j].copy()
            if f.nbytes > self._room[0]:
                self._room[0] = 0
                return
            self._room[0] -= f.nbytes
            self._cache[chunk[i][0]] = (f, int(lengths[i]))


def bucketed_order(items, batch_size, rng=None):
    """Length-bucketed batching: items [audio, label, duration] sorted by duration, cut into
    mini-batches, and the mini-batches -- not the items -- shuffled.  Every batch then holds utterances of
    similar length, so the recurrence (which stops at the batch's longest utterance) wastes no frames."""
    import random
    ordered = sorted(items, key=lambda it: (it[2] if len(it) > 2 and it[2] is not None else 0.0))
    groups = [ordered[i:i + batch_size] for i in range(0, len(ordered), batch_size)]
    tail = [groups.pop()] if groups and len(groups[-1]) < batch_size else []     # a short group stays last,
    (rng or random).shuffle(groups)                                              # or every later batch straddles
    return [it for g in groups + tail for it in g]


class DatasetIterator(object):
    """Iterator over a dataset's mini-batches.  `prefetch()` prepares the NEXT mini-batch ahead of its use: the
    host part comes from the dataset's decode-ahead thread, the device part (upload, resampling, front-end
    kernels) is enqueued on a side stream, so it runs under the tail of the training step in flight (the
    reference's tf.data pipeline overlaps feature extraction with training the same way, :820-822)."""

    _UNSET = object()

    def __init__(self, dataset):
        self._dataset = dataset
        self._gen = None
        self._ahead = self._UNSET
        self._side = None
        self.initializer = _Op(self._reset)

    def _reset(self):
        self._gen = self._dataset.batches()
        self._ahead = self._UNSET

    @property
    def dataset(self):
        return self._dataset

    def make_initializer(self, dataset):
        def _swap():
            self._dataset = dataset
            self._reset()
        return _Op(_swap)

    def prefetch(self, after=None):
        """Prepare the next mini-batch ahead of its use.  after: an event the device half must wait for.
        Returns the event that marks the device half done (None when there is nothing new in flight)."""
        if self._gen is None or self._ahead is not self._UNSET:
            return None
        if not torch.cuda.is_available():        # host-only runs (multi-process CPU tests): nothing to overlap
            try:
                self._ahead = (next(self._gen), None)
            except StopIteration:
                self._ahead = None
            return None
        if self._side is None:
            self._side = torch.cuda.Stream()
        if after is not None:
            self._side.wait_event(after)
        try:
            with torch.cuda.stream(self._side):
                batch = next(self._gen)
                done = torch.cuda.Event()
                done.record(self._side)
            self._ahead = (batch, done)
            return done
        except StopIteration:
            self._ahead = None
            return None

    def has_next(self):
        """True when another mini-batch is available (prepares it ahead if that has not happened yet)."""
        if self._gen is None:
            return False
        self.prefetch()
        return self._ahead is not None

    def get_next(self):
        if self._gen is None:
            raise RuntimeError("iterator used before its initializer was run")
        self.prefetch()
        ahead, self._ahead = self._ahead, self._UNSET
        if ahead is None:
            self._ahead = None
            raise OutOfRangeError()
        batch, done = ahead
        if done is not None:
            cur = torch.cuda.current_stream()
            cur.wait_event(done)
            if torch.is_tensor(batch[0]):
                batch[0].record_stream(cur)
        return batch


def _edit_distance(a, b):
    """Levenshtein distance between two integer sequences, one numpy pass per row:
    cur[j] = min(sub/del candidates c[j], cur[j-1] + 1) is a prefix minimum of c[k] - k."""
    a = np.asarray(a, np.int64)
    b = np.asarray(b, np.int64)
    if len(a) == 0 or len(b) == 0:
        return int(max(len(a), len(b)))
    ramp = np.arange(len(b) + 1)
    prev = ramp.copy()
    for i in range(1, len(a) + 1):
        cand = np.empty(len(b) + 1, np.int64)
        cand[0] = i
        cand[1:] = np.minimum(prev[:-1] + (b != a[i - 1]), prev[1:] + 1)
        prev = np.minimum.accumulate(cand - ramp) + ramp
    return int(prev[-1])


# ------------------------------------------------------------------------ model
def _engine_stream(fn):
    """Run the method on the engine's own non-default stream (see Engine.on_stream)."""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *a, **kw):
        if getattr(self, "engine", None) is None:
            return fn(self, *a, **kw)
        with self.engine.on_stream():
            return fn(self, *a, **kw)
    return wrapped


class _AsyncBeamDecoder(object):
    """The reference decodes EVERY training mini-batch with the width-100 beam decoder and takes the logged error rate -- which
    drives the learning-rate plateau rule of stt.py:219-231 -- from that (models/AcousticModel.py:312-314, :370, :641).  Here
    the decode leaves the training thread: the logits of a mini-batch are copied to pinned host memory by a DMA on a side
    stream as soon as they exist (behind the output layer: the copy engine works beside the CTC stage and the backward
    recurrence, it needs no compute unit), a host thread waits for that copy, runs csrc/beam.cpp (one thread per utterance
    inside the library) and the host edit distance, and the training thread collects the result up to `lag` mini-batches
    later.  lag = 0 collects at once: the reference's timing exactly, at the price of waiting for the decoder."""

    def __init__(self, engine, beam_width, merge_repeated, lag):
        from concurrent.futures import ThreadPoolExecutor
        self.engine, self.beam_width, self.merge_repeated, self.lag = engine, int(beam_width), bool(merge_repeated), max(0, int(lag))
        # decode threads per mini-batch: with results due `lag` steps later a job may take that long, and a steady load on a few
        # cores disturbs the training thread (and a container's CPU quota) less than a burst of one thread per utterance;
        # 16 measured best on a 16-core quota -- but never more than this rank's
        # share of the host (LOCAL_WORLD_SIZE ranks per node, two cores left for the training and prefetch threads): eight
        # ranks x 16 threads on a 64-core host would stall every rank's collect()
        share = max(1, (os.cpu_count() or 16) // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1"))) - 2)
        self.threads = int(os.environ.get("AMDSPEECH_TRAIN_DECODER_THREADS", "0")) or (0 if self.lag == 0 else min(16, share))
        T, B, C = engine.logits.shape
        self._free = [torch.empty(T, B, C, dtype=torch.float32).pin_memory() for _ in range(self.lag + 2)]
        self._copy_stream = torch.cuda.Stream(device=engine.device)
        self._pool = ThreadPoolExecutor(max_workers=self.lag + 1, thread_name_prefix="amdspeech-beam")
        self._pending = collections.deque()
        self._results_ready = []
        self._last_copy = None

    def submit(self, after_event, Tr, lengths, dense, num_labels):
        """Called where the logits of this mini-batch are complete on the engine's stream (`after_event`)."""
        while not self._free:                     # (every buffer in flight: collect the oldest first)
            self._results_ready.append(self._collect_one())
        buf = self._free.pop()
        self._copy_stream.wait_event(after_event)
        with torch.cuda.stream(self._copy_stream):
            buf[:Tr].copy_(self.engine.logits[:Tr], non_blocking=True)
            done = torch.cuda.Event()
            done.record(self._copy_stream)
        self._last_copy = done
        lens = np.minimum(np.asarray(lengths, np.int32), Tr).astype(np.int32)
        truth_rows = []
        for row in np.asarray(dense):
            kept = row[row != 0]
            truth_rows.append(kept if len(kept) else np.array([num_labels - 1], np.int32))      # (:155-159)
        fut = self._pool.submit(self._decode, buf, done, Tr, lens, truth_rows, num_labels)
        self._pending.append((fut, buf))

    def _decode(self, buf, done, Tr, lens, truth_rows, num_labels):
        while not done.query():              # (not done.synchronize(): that spins on a core for the milliseconds the copy
            time.sleep(0.0002)               #  waits behind the backward kernel, which leaves it no compute unit)
        ids, out_len, _ = ops.ctc_beam_search(buf[:Tr].numpy(), lens, self.beam_width, self.merge_repeated,
                                              max_threads=self.threads)
        B = len(truth_rows)
        width = max(max(len(r) for r in truth_rows), 1)
        truth = np.zeros((B, width), np.int32)
        tlen = np.zeros(B, np.int32)
        for b, r in enumerate(truth_rows):
            truth[b, :len(r)] = r
            tlen[b] = len(r)
        dist = ops.edit_distance_host(ids, out_len, truth, tlen)
        return float(np.mean(dist / tlen.astype(np.float64)))       # mean of edit_distance / len(truth) (:370)

    def _collect_one(self):
        fut, buf = self._pending.popleft()
        val = fut.result()
        self._free.append(buf)
        return val

    def collect(self, drain=False, at_least_one=False):
        """Error rates of the mini-batches that are due: everything beyond `lag` in flight (all of them when draining; the
        oldest one, waited for, when the caller needs a value and none is due)."""
        out = list(self._results_ready)
        self._results_ready = []
        while self._pending and (drain or len(self._pending) > self.lag or (at_least_one and not out)):
            out.append(self._collect_one())
        return out

    def discard_last(self):
        """The mini-batch submitted last turned out invalid (a dataflow time-out): its decode is waited for and thrown away."""
        if self._pending:
            fut, buf = self._pending.pop()
            try:
                fut.result()
            except Exception:      # (garbage logits: whatever the decoder made of them)
                pass
            self._free.append(buf)

    def guard(self, stream):
        """The next forward pass overwrites the logits: it has to stay behind the last copy (long finished in practice)."""
        if self._last_copy is not None:
            stream.wait_event(self._last_copy)

    def close(self):
        self.collect(drain=True)
        self._pool.shutdown(wait=True)


class AcousticModel(object):
    def __init__(self, num_layers, hidden_size, batch_size, max_input_seq_length,
                 max_target_seq_length, input_dim, normalization, num_labels):
        self.num_layers = num_layers
        self.hidden_size = hidden_size
        self.batch_size = batch_size
        self.max_input_seq_length = max_input_seq_length
        self.max_target_seq_length = max_target_seq_length
        self.input_dim = input_dim
        self.normalization = normalization
        self.num_labels = num_labels
        self.engine = None
        self.rnn_created = False
        self.forward_only = True
        self.input_keep_prob = self.output_keep_prob = 1.0
        self.grad_clip = 1.0
        self.lr_decay_factor = 1.0
        self.learning_rate_var = _Variable(0.0)
        self.learning_rate_decay_op = _Op(self._decay_lr)
        self.global_step = _Variable(0)
        self.is_training = False
        self.tensorboard_dir = None
        self.timeline_enabled = False
        self.compute_error_rate = True     # the reference decodes on every mini-batch (:641)
        # Decoder behind `prediction` (:312-314).  Inference and evaluation (process_input, evaluate_full) use what
        # the reference uses: prefix beam search of width 100 (TensorFlow's default) + merge_repeated, on host
        # threads (csrc/beam.cpp).  The per-mini-batch TRAINING error rate -- a logging scalar the reference also
        # derives from the beam decoder (:641) -- is decoded greedily on the GPU (10^7 prefix extensions
        # per utterance per step would cost several optimiser steps of host time); set train_decoder = "beam" to
        # reproduce the reference exactly there too.
        self.decoder = "beam"
        self.train_decoder = "greedy"
        # train_decoder = "beam": the reference's decoder on every training mini-batch, asynchronously (_AsyncBeamDecoder); the
        # error rate a step reports is then that of the mini-batch `train_decoder_lag` mini-batches earlier (0: no lag, the
        # training thread waits for the decoder).  Evaluation passes always decode synchronously.
        self.train_decoder_lag = 1
        self._async_beam = None
        self._drain_decoder = False
        self._grads_kept = None          # (run_step: the gradient buffer before a mini-batch that accumulates into it)
        self.recovered_steps = 0         # mini-batches repeated on the launch-per-diagonal kernels after a dataflow time-out
        self._step_invalid = False       # ... and the repeat failed too: end_batch skips the optimiser step (on every rank)
        self.skipped_steps = 0
        self._skipped_in_a_row = 0       # ... consecutively: a persistent fault must not become a silent no-progress loop
        self.max_skipped_in_a_row = 8
        self._err_batches = 0
        self._last_err = None
        self.precision = "f32"             # "bf16x3": opt-in split-precision MFMA in the recurrence (config key `precision`)
        self.bidirectional = False         # config key `bidirectional` (BASELINE configs[4]; the reference is unidirectional)
        self.sync_batch_norm = False       # config key `sync_batch_norm`: data-parallel batch-norm moments over ALL ranks (deviation)
        self.save_tf_bundle = False        # also write <stem>.index / .data-00000-of-00001 on save()
        self.save_optimizer_state = True   # native .npz also carries Adam m/v/step and the RNN state
        self.beam_width = 100
        self.merge_repeated = True
        self._train_iter = self._valid_iter = self._single_iter = None
        self._acc_loss = self._acc_err = 0.0
        self._mini_batches = 0
        self._dropout_seed = 0
        self._placeholder_batch = None
        self._next_agreed = None           # data parallel: "every rank has another mini-batch", agreed one step ahead

    # ---- graph construction ----------------------------------------------------
    def _make_engine(self):
        if self.rnn_created:
            logging.fatal("Trying to create the acoustic RNN but it is already.")
        self.engine = Engine(self.num_layers, self.hidden_size, self.input_dim, self.num_labels,
                             self.batch_size, self.max_input_seq_length, self.max_target_seq_length,
                             normalization=bool(self.normalization), precision=self.precision,
                             bidirectional=bool(self.bidirectional), sync_batch_norm=bool(self.sync_batch_norm))
        self.rnn_created = True

    def create_forward_rnn(self):
        self._make_engine()
        self.forward_only = True
        return self.engine.logits

    def create_training_rnn(self, input_keep_prob, output_keep_prob, grad_clip, learning_rate,
                            lr_decay_factor, use_iterator=False):
        self._make_engine()
        self.forward_only = False
        self.input_keep_prob, self.output_keep_prob = float(input_keep_prob), float(output_keep_prob)
        self.grad_clip = float(grad_clip)
        self.lr_decay_factor = float(lr_decay_factor)
        self.learning_rate_var.value = float(learning_rate)
        self.use_iterator = use_iterator

    def _decay_lr(self):
        self.learning_rate_var.value *= self.lr_decay_factor
        return self.learning_rate_var.value

    def add_tensorboard(self, session, tensorboard_dir, tb_run_name=None, timeline_enabled=False):
        """TensorBoard summaries are out of scope; with `timeline_enabled` (stt.py --timeline) the directory receives the
        reference's `timeline-<action>.ctf.json` files (:873-885) in chrome trace format, one per action of the last
        optimiser step (`step-i`, `end_batch`): stage times from HIP events instead of TensorFlow's per-op step stats."""
        self.tensorboard_dir = tensorboard_dir
        self.timeline_enabled = timeline_enabled

    def _write_timeline(self, action, spans, host_start):
        """spans: [(name, start_ms, dur_ms)] on the GPU stream; plus one host span for the whole action."""
        if not self.timeline_enabled or dataparallel.current().rank != 0:      # (one writer per job)
            return
        if self.tensorboard_dir is None:
            logging.warning("Could not write timeline, a tensorboard_dir is required in config file")
            return
        import json
        os.makedirs(self.tensorboard_dir, exist_ok=True)
        events = [{"name": "process_name", "ph": "M", "pid": 0, "args": {"name": "MI355X stream"}},
                  {"name": "process_name", "ph": "M", "pid": 1, "args": {"name": "host"}},
                  {"name": action, "ph": "X", "pid": 1, "tid": 0, "ts": 0.0, "dur": (time.time() - host_start) * 1e6}]
        for name, start_ms, dur_ms in spans:
            events.append({"name": name, "ph": "X", "pid": 0, "tid": 0, "ts": start_ms * 1e3, "dur": dur_ms * 1e3})
        path = os.path.join(self.tensorboard_dir, "timeline-" + action + ".ctf.json")
        logging.info("Writing to %s", os.path.basename(path))
        with open(path, "w") as f:
            json.dump({"traceEvents": events, "displayTimeUnit": "ms"}, f)

    def get_learning_rate(self):
        return self.learning_rate_var.value

    def set_learning_rate(self, sess, learning_rate):
        self.learning_rate_var.value = float(learning_rate)

    def set_is_training(self, sess, is_training):
        self.is_training = bool(is_training)

    @staticmethod
    def initialize(sess):
        return None   # parameters are initialised when the engine is built

    # ---- checkpoints -------------------------------------------------------------
    _TF_NAMES = {"input_w": "Input_Layer/input_w", "input_b": "Input_Layer/input_b",
                 "output_w": "Output_layer/output_w", "output_b": "Output_layer/output_b"}

    def _tf_name(self, name):
        if name in self._TF_NAMES:
            return self._TF_NAMES[name]
        if name.startswith("bw_"):       # (bidirectional build only: tf.nn.bidirectional_dynamic_rnn's scope names)
            kind, l = name[3:].split("_")
            return "bidirectional_rnn/bw/multi_rnn_cell/cell_%s/basic_lstm_cell/%s" % (l, kind)
        kind, l = name.split("_")
        return "rnn/multi_rnn_cell/cell_%s/basic_lstm_cell/%s" % (l, kind)

    def save(self, session, checkpoint_dir):
        """The reference's Saver writes weights, biases, global_step and learning_rate (:518-522) -- no Adam slots,
        no RNN state, so its resumed runs restart Adam cold.  The native .npz keeps those names (a reference
        checkpoint maps 1:1) and ADDS the optimiser moments, the Adam step count and the persistent RNN state under
        `adam/...` / `rnn_state/...` keys.  Data parallel: rank 0 writes, everybody waits."""
        grp = dataparallel.current()
        if grp.rank == 0:
            os.makedirs(checkpoint_dir, exist_ok=True)
            eng = self.engine
            arrays = {self._tf_name(k): v for k, v in eng.to_numpy().items()}
            arrays["global_step"] = np.int32(self.global_step.value)
            arrays["learning_rate"] = np.float32(self.learning_rate_var.value)
            stem = "acousticmodel.ckpt-%d" % self.global_step.value
            extra = {}
            if self.save_optimizer_state:
                for slot, flat in (("m", eng.adam_m), ("v", eng.adam_v)):
                    for k, v in eng.to_numpy(flat).items():
                        extra["adam/%s/%s" % (slot, self._tf_name(k))] = v
                extra["adam/step"] = np.int64(eng.adam_step)
                extra["rnn_state/h"] = eng.state_h.cpu().numpy()
                extra["rnn_state/c"] = eng.state_c.cpu().numpy()
            tmp = os.path.join(checkpoint_dir, stem + ".tmp.npz")
            np.savez(tmp, **arrays, **extra)
            os.replace(tmp, os.path.join(checkpoint_dir, stem + ".npz"))      # never a half-written checkpoint
            if self.save_tf_bundle:      # additionally the TensorFlow bundle a reference tf.train.Saver can restore
                from . import tf_bundle
                tf_bundle.write_bundle(os.path.join(checkpoint_dir, stem), arrays)
            with open(os.path.join(checkpoint_dir, "checkpoint"), "w") as fh:
                fh.write('model_checkpoint_path: "%s"\n' % stem)
            logging.info("Checkpoint saved")
        grp.barrier()

    def restore(self, session, checkpoint_dir):
        grp = dataparallel.current()
        marker = os.path.join(checkpoint_dir, "checkpoint")
        found = grp.broadcast_object(os.path.exists(marker))      # rank 0's view decides for everyone
        if not found:
            logging.info("Created model with fresh parameters.")
            if grp.world > 1:
                self.engine.broadcast_state(0)
            return
        eng = self.engine
        failure = None
        if grp.rank == 0:            # ONLY rank 0 touches the checkpoint directory (it need not be shared between nodes)
            try:
                with open(marker) as fh:
                    stem = fh.read().split('"')[1]
                logging.info("Reading model parameters from %s", stem)
                npz = os.path.join(checkpoint_dir, stem + ".npz")
                if os.path.exists(npz):
                    z = np.load(npz)
                else:                        # a TensorFlow bundle written by the reference (:483-487)
                    from . import tf_bundle
                    z = tf_bundle.read_bundle(os.path.join(checkpoint_dir, stem))
                names = eng.layout.names()
                eng.load_numpy({k: z[self._tf_name(k)] for k in names})
                self.global_step.value = int(z["global_step"])
                self.learning_rate_var.value = float(z["learning_rate"])
                keys = set(z.keys()) if hasattr(z, "keys") else set(z)
                if "adam/step" in keys:      # native checkpoint: resume the optimiser warm
                    eng.load_numpy({k: z["adam/m/" + self._tf_name(k)] for k in names}, flat=eng.adam_m)
                    eng.load_numpy({k: z["adam/v/" + self._tf_name(k)] for k in names}, flat=eng.adam_v)
                    eng.adam_step = int(z["adam/step"])
                    if tuple(z["rnn_state/h"].shape) == tuple(eng.state_h.shape):      # (batch size may have changed)
                        eng.state_h.copy_(torch.as_tensor(z["rnn_state/h"]))
                        eng.state_c.copy_(torch.as_tensor(z["rnn_state/c"]))
                else:                        # reference-style checkpoint: Adam restarts cold, like the reference
                    eng.adam_m.zero_(); eng.adam_v.zero_(); eng.adam_step = 0
            except Exception as exc:      # missing / corrupt file, key or shape mismatch ...
                failure = "%s: %s" % (type(exc).__name__, exc)
        # the other ranks are about to park in a broadcast: they must hear about a failed read FIRST, or the job hangs
        # instead of failing (every rank used to read the checkpoint itself and fail on its own)
        failure = grp.broadcast_object(failure)
        if failure is not None:
            raise RuntimeError("restore: rank 0 could not read the checkpoint in %s (%s)" % (checkpoint_dir, failure))
        if grp.world > 1:            # replicas start bit-identical to what rank 0 read
            eng.broadcast_state(0)
            grp.broadcast_(eng.state_h.view(-1), 0)
            grp.broadcast_(eng.state_c.view(-1), 0)
            self.global_step.value = int(grp.broadcast_object(self.global_step.value))
            self.learning_rate_var.value = float(grp.broadcast_object(self.learning_rate_var.value))

    # ---- metrics ---------------------------------------------------------------
    @staticmethod
    def calculate_wer(first_string, second_string):
        return _edit_distance_tokens(first_string.split(), second_string.split())

    @staticmethod
    def calculate_cer(first_string, second_string):
        return _edit_distance_tokens(list(first_string.replace(" ", "")), list(second_string.replace(" ", "")))

    # ---- input plumbing ----------------------------------------------------------
    @staticmethod
    def build_dataset(input_set, batch_size, max_input_seq_length, max_target_seq_length,
                      signal_processing, char_map, n_mfcc=20, prefetch=2, feature_cache_mb=0, sample_rate=22050):
        return AcousticDataset(input_set, batch_size, max_input_seq_length, max_target_seq_length,
                               signal_processing, char_map, n_mfcc=n_mfcc, prefetch=prefetch,
                               feature_cache_mb=feature_cache_mb, sample_rate=sample_rate)

    def add_dataset_input(self, dataset):
        self._single_iter = DatasetIterator(dataset)
        return self._single_iter

    def add_datasets_input(self, train_dataset, valid_dataset):
        self._train_iter, self._valid_iter = DatasetIterator(train_dataset), DatasetIterator(valid_dataset)
        return self._train_iter, self._valid_iter

    def feed(self, inputs, input_seq_lengths, labels):
        """Placeholder path (use_iterator=False): what feeding inputs_ph / input_seq_lengths_ph /
        labels_ph does in the reference."""
        self._placeholder_batch = (inputs, np.asarray(input_seq_lengths, np.int32), np.asarray(labels, np.int32))

    def _prefetch_next(self, after=None):
        if self._placeholder_batch is not None:
            return None
        it = self._single_iter
        if it is None:
            it = self._train_iter if self.is_training else self._valid_iter
        if it is not None:
            return it.prefetch(after)
        return None

    def _has_next(self):
        if self._placeholder_batch is not None:
            return True
        it = self._single_iter
        if it is None:
            it = self._train_iter if self.is_training else self._valid_iter
        return it is not None and it.has_next()

    def _next_batch(self):
        if self._placeholder_batch is not None:
            b, self._placeholder_batch = self._placeholder_batch, None
            return b
        it = self._single_iter
        if it is None:
            it = self._train_iter if self.is_training else self._valid_iter
        if it is None:
            raise RuntimeError("no input: add a dataset or feed() a batch first")
        return it.get_next()

    @staticmethod
    def _host_max(lengths):
        """Longest utterance of the batch when the lengths are still on the host (they are in the
        dataset / placeholder paths): lets the engine stop the recurrence there, as dynamic_rnn does."""
        if torch.is_tensor(lengths):
            return None if lengths.is_cuda else int(lengths.max())
        a = np.asarray(lengths)
        return int(a.max()) if a.size else None

    def _to_device(self, inputs, lengths, dense):
        dev = self.engine.device
        x = inputs if torch.is_tensor(inputs) else torch.as_tensor(np.asarray(inputs, np.float32))
        x = x.to(dev, torch.float32).contiguous()
        return x, torch.as_tensor(lengths, dtype=torch.int32).to(dev), torch.as_tensor(dense, dtype=torch.int32).to(dev)

    # ---- step orchestration (:634-703, :887-939) -----------------------------------
    @_engine_stream
    def start_batch(self, session, is_training, run_options=None, run_metadata=None):
        self._acc_loss = self._acc_err = 0.0
        self._mini_batches = self._err_batches = self._loss_batches = 0
        self._step_invalid = False
        self.set_is_training(session, is_training)
        if is_training:
            self.engine.zero_grads()

    @_engine_stream
    def run_step(self, session, compute_gradients=True, run_options=None, run_metadata=None):
        start = time.time()
        inputs, lengths, dense = self._next_batch()          # may raise OutOfRangeError
        x, dlen, dlab = self._to_device(inputs, lengths, dense)
        eng = self.engine
        keep = (self.input_keep_prob, self.output_keep_prob) if compute_gradients else (1.0, 1.0)
        self._dropout_seed += 1
        marks = [] if self.timeline_enabled else None
        use_async = self.compute_error_rate and self.train_decoder == "beam" and compute_gradients
        if self._async_beam is not None:                      # (this forward pass overwrites the logits a copy may still read)
            self._async_beam.guard(torch.cuda.current_stream(eng.device))
        decode_hook = None
        if use_async:
            if self._async_beam is None:
                self._async_beam = _AsyncBeamDecoder(eng, self.beam_width, self.merge_repeated, self.train_decoder_lag)

            def decode_hook(after, _self=self, _lengths=lengths, _dense=dense):
                # (the logits of THIS mini-batch exist behind `after`: their way to the host starts beside the CTC stage)
                _self._async_beam.submit(after, eng._Tr, _lengths, _dense, _self.num_labels)
                return None
        # the next batch's upload + front end need nothing of this step: beside the forward recurrence where that leaves XCDs
        # idle, else beside the CTC stage (Engine.mini_batch)
        if compute_gradients and self._mini_batches > 0:
            # gradients of earlier mini-batches of this optimiser step are in the buffer: what a time-out of THIS mini-batch must
            # not take with it (a 25 MB device copy, ~10 us; the first mini-batch of a step starts from zeros: nothing to keep)
            if self._grads_kept is None:
                self._grads_kept = torch.empty_like(eng.grads)
            self._grads_kept.copy_(eng.grads)
        eng.mini_batch(x, dlen, dlab, keep[0], keep[1], seed=self._dropout_seed, use_state=True,
                       compute_gradients=compute_gradients, max_len=self._host_max(lengths),
                       beside_ctc=decode_hook, beside_forward=self._prefetch_next, marks=marks)
        grp = dataparallel.current()
        if grp.world > 1 and compute_gradients:
            # data parallel: agree NOW (host channel, while the GPU works on this step) whether every rank has
            # another mini-batch, so that no rank ever enters a gradient all-reduce the others skip
            self._next_agreed = grp.all_true(self._has_next())
        # the error rate's kernels go out BEFORE the loss is read back: one drain of the stream covers both read-backs
        pending_err = self._error_rate_launch(dlen, dense) if (self.compute_error_rate and not use_async) else None
        loss = eng.loss.cpu().numpy().astype(np.float64)
        mini_batch_valid = True                               # (an invalid mini-batch adds nothing to the logged loss / error rate)
        if not eng.healthy():                                 # (the stream is drained by the read-back above)
            # A whole-sequence launch of this mini-batch gave up waiting (its workgroups were not all resident: another process
            # on the GPU, a tool that serialises kernels).  The reference's loop never loses a step (:887-939): take the
            # mini-batch's gradient contribution back, run it again on the launch-per-diagonal kernels -- same inputs, same
            # dropout seed, the persistent RNN state not yet touched -- and go on; the first time is logged.
            self.recovered_steps += 1
            if self.recovered_steps == 1:
                logging.warning("a whole-sequence LSTM launch timed out (mini-batch %d of step %d): repeating it on the "
                                "launch-per-diagonal kernels; further time-outs are counted in recovered_steps",
                                self._mini_batches, self.global_step.value)
            if compute_gradients:
                if self._mini_batches > 0:
                    eng.grads.copy_(self._grads_kept)
                else:
                    eng.zero_grads()
            if use_async:
                self._async_beam.discard_last()               # (the copy of the invalid logits)
                self._async_beam.guard(torch.cuda.current_stream(eng.device))
            if getattr(eng, "sync_batch_norm", False) and grp.world > 1:
                # Batch norm over the GLOBAL batch: forward and backward of a mini-batch are cross-rank collectives
                # (ops.batchnorm_fwd_dp / _bwd_dp).  A repeat on THIS rank alone would enter collectives its peers never match --
                # they would pair with the peers' next mini-batch or with the gradient all-reduce.  No local repeat: the
                # mini-batch is dropped and end_batch drops the optimiser step on every rank (the agreement over the host group).
                mini_batch_valid = False
                self._step_invalid = True
                pending_err = None
                logging.error("a whole-sequence launch timed out under sync_batch_norm: optimiser step %d will be skipped on "
                              "every rank", self.global_step.value)
            else:
                eng.mini_batch(x, dlen, dlab, keep[0], keep[1], seed=self._dropout_seed, use_state=True,
                               compute_gradients=compute_gradients, max_len=self._host_max(lengths),
                               beside_ctc=decode_hook, per_diagonal=True)
                pending_err = self._error_rate_launch(dlen, dense) if (self.compute_error_rate and not use_async) else None
                loss = eng.loss.cpu().numpy().astype(np.float64)
                if not eng.healthy() or not np.all(np.isfinite(loss[np.asarray(lengths) > 0])):
                    mini_batch_valid = False
                    self._step_invalid = True                 # (end_batch: no rank applies this optimiser step)
                    pending_err = None
                    logging.error("the repeated mini-batch is invalid too: optimiser step %d will be skipped", self.global_step.value)
        if mini_batch_valid:
            eng.keep_state()                                  # rnn_keep_state_op, fetched on every step (:642)
            with np.errstate(divide="ignore", invalid="ignore"):
                self._acc_loss += float(np.mean(loss / np.asarray(lengths, np.float64)))   # :361
            self._loss_batches += 1
        if pending_err is not None:
            dist, tlen = pending_err
            self._acc_err += float(np.mean(dist.cpu().numpy() / tlen.astype(np.float64)))
            self._err_batches += 1
        elif use_async:
            for val in self._async_beam.collect():            # (the decodes that are due: `train_decoder_lag` mini-batches old)
                self._acc_err += val
                self._err_batches += 1
        self._mini_batches += 1
        if marks:
            torch.cuda.synchronize()
            t0 = marks[0][1]
            spans = [(name, t0.elapsed_time(prev), prev.elapsed_time(ev))
                     for (_, prev), (name, ev) in zip(marks[:-1], marks[1:])]
            self._write_timeline("step-%d" % (self._mini_batches - 1), spans, start)
        logging.debug("Step duration : %.2f", time.time() - start)
        return self._mini_batches

    def _error_rate(self, dlen, dense):
        dist, tlen = self._error_rate_launch(dlen, dense)
        return float(np.mean(dist.cpu().numpy() / tlen.astype(np.float64)))

    def _error_rate_launch(self, dlen, dense):
        """mean over the batch of edit_distance(prediction, truth) / len(truth) (:370); truth keeps the
        EOS token, drops id 0, and empty rows are [C-1] (:155-159).  Decode, merge and distance run on
        the GPU; returns (device distances [B], host truth lengths): one 4*B-byte copy comes back later."""
        if self.train_decoder == "beam":
            hid, hlen, _ = ops.ctc_beam_search(self.engine.logits, dlen, self.beam_width, self.merge_repeated)
            dev0 = self.engine.device
            ids, out_len = torch.as_tensor(hid).to(dev0), torch.as_tensor(hlen).to(dev0)
        else:
            ids, out_len = ops.ctc_greedy_decode(self.engine.logits, dlen, ws=self.engine.ctc_ws)
            if self.merge_repeated:
                ops.merge_repeated(ids, out_len, self.num_labels)
        truth = np.zeros_like(dense)
        tlen = np.zeros(self.batch_size, np.int32)
        for b in range(self.batch_size):
            kept = dense[b][dense[b] != 0]
            if len(kept) == 0:
                kept = np.array([self.num_labels - 1], np.int32)
            truth[b, :len(kept)] = kept
            tlen[b] = len(kept)
        dev = self.engine.device
        dist = ops.edit_distance(ids, out_len, torch.as_tensor(truth).to(dev), torch.as_tensor(tlen).to(dev))
        return dist, tlen

    @_engine_stream
    def end_batch(self, session, is_training, run_options=None, run_metadata=None, rnn_state_reset_ratio=1.0):
        tl_start, tl_marks = time.time(), None
        if is_training:
            if self.timeline_enabled and torch.cuda.is_available():
                tl_marks = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                tl_marks[0].record(torch.cuda.current_stream())
            # A rank whose mini-batch stayed invalid even on the launch-per-diagonal kernels has no gradients to give.  Data parallel:
            # the ranks AGREE over the host group before anybody enters the all-reduce (one int over gloo, ~0.1 ms per optimiser
            # step) -- either every rank exchanges and applies, or none does and the step is dropped everywhere (replicas stay
            # identical, nobody waits in a collective the others skip)
            grp = dataparallel.current()
            valid = grp.all_true(not self._step_invalid) if grp.world > 1 else not self._step_invalid
            if valid:
                self.engine.all_reduce_grads()
                if tl_marks:
                    tl_marks[1].record(torch.cuda.current_stream())
                self.engine.apply(self.learning_rate_var.value, self.grad_clip)
                if tl_marks:
                    tl_marks[2].record(torch.cuda.current_stream())
                    torch.cuda.synchronize()
                    a, b, c = tl_marks
                    self._write_timeline("end_batch", [("gradient all-reduce", 0.0, a.elapsed_time(b)),
                                                       ("clip + Adam", a.elapsed_time(b), b.elapsed_time(c))], tl_start)
                self.global_step.value += 1
                self._skipped_in_a_row = 0
            else:
                self.skipped_steps += 1
                self._skipped_in_a_row += 1
                self.engine.zero_grads()
                logging.error("optimiser step dropped on every rank: a mini-batch of it was invalid on %s",
                              "this rank" if self._step_invalid else "another rank")
                if self._skipped_in_a_row >= self.max_skipped_in_a_row:
                    # (every rank counts the same agreed outcome, so every rank raises: no one is left in a collective)
                    raise RuntimeError("%d optimiser steps in a row were dropped: the device does not run the LSTM kernels "
                                       "to completion (see the log) -- giving up instead of looping without progress"
                                       % self._skipped_in_a_row)
            if randint(1, int(1 // rnn_state_reset_ratio)) == 1:
                self.engine.zero_state()
        if is_training and self._async_beam is not None and self.compute_error_rate:
            if self._drain_decoder:
                # the epoch (or the run) ends with this step: the decodes still in flight belong to it -- waiting for them here
                # is the only way they ever reach a reported error rate
                self._drain_decoder = False
                for val in self._async_beam.collect(drain=True):
                    self._acc_err += val
                    self._err_batches += 1
            if self._err_batches == 0:
                # nothing was due in this step (the first `lag` steps of a run): the very first step waits for its own decode,
                # after that a step without a new value reports the latest one again
                if self._last_err is None:
                    for val in self._async_beam.collect(at_least_one=True):
                        self._acc_err += val
                        self._err_batches += 1
                else:
                    self._acc_err, self._err_batches = self._last_err, 1
            self._last_err = self._acc_err / max(self._err_batches, 1)
        loss_sum, err_sum, n, n_err = self._acc_loss, self._acc_err, float(getattr(self, "_loss_batches", self._mini_batches)), float(self._err_batches)
        if is_training:
            # the logging scalars are summed over the ranks: every rank reports -- and feeds to the
            # learning-rate plateau rule of stt.py -- the SAME mean loss / error rate
            loss_sum, err_sum, n, n_err = dataparallel.current().sum_scalars([loss_sum, err_sum, n, n_err])
        n = max(n, 1.0)
        return loss_sum / n, err_sum / max(n_err, 1.0), self.global_step.value

    def close(self):
        """End of the model's life (stt.py calls it when training / evaluation is over): waits for the asynchronous training
        decoder's host threads -- none may sit in event.synchronize() while the interpreter or the HIP runtime shuts down --
        and gives its pinned buffers back.  Idempotent; also run by __del__."""
        dec, self._async_beam = getattr(self, "_async_beam", None), None
        if dec is not None:
            try:
                dec.close()
            except Exception:      # (interpreter shutdown: nothing left to protect)
                pass

    def __del__(self):
        self.close()

    def run_train_step(self, sess, mini_batch_size, rnn_state_reset_ratio, run_options=None, run_metadata=None):
        start_time = time.time()
        dataset_empty = False
        self.start_batch(sess, True)
        mini_batch_num = 0
        grp = dataparallel.current()
        try:
            for _ in range(mini_batch_size):
                if grp.world > 1:
                    # every rank leaves the epoch together: a rank whose shard still had a mini-batch drops it
                    # (dataparallel.shard gives equal shards, so this only guards against unequal inputs)
                    agreed, self._next_agreed = self._next_agreed, None
                    if agreed is None:
                        self.set_is_training(sess, True)
                        agreed = grp.all_true(self._has_next())
                    if not agreed:
                        raise OutOfRangeError()
                mini_batch_num = self.run_step(sess, True)
        except OutOfRangeError:
            logging.debug("Dataset empty, exiting train step")
            dataset_empty = True
            self._drain_decoder = True
        if mini_batch_num > 0:
            mean_loss, mean_error_rate, current_step = self.end_batch(
                sess, True, rnn_state_reset_ratio=rnn_state_reset_ratio)
            logging.info("Batch %d : loss %.5f - error_rate %.5f - duration %.2f",
                         current_step, mean_loss, mean_error_rate, time.time() - start_time)
            return mean_loss, mean_error_rate, current_step, dataset_empty
        if dataset_empty and self._drain_decoder:
            # the epoch ended exactly on a step boundary: no end_batch runs for this call, so the previous step's decodes still in
            # flight are waited for HERE and folded into the latest error rate -- left alone they would surface in the first step
            # of the NEXT epoch (or be dropped by close()), i.e. feed the learning-rate plateau rule from the wrong window
            self._drain_decoder = False
            if self._async_beam is not None and self.compute_error_rate:
                late = list(self._async_beam.collect(drain=True))
                if late:
                    self._last_err = sum(late) / len(late)
        return 0.0, 0.0, self.global_step.value, dataset_empty

    def run_evaluation(self, sess, run_options=None, run_metadata=None):
        start_time = time.time()
        logging.info("Start evaluating...")
        self.start_batch(sess, False)
        try:
            while True:
                self.run_step(sess, False)
        except OutOfRangeError:
            logging.debug("Dataset empty, exiting evaluation step")
        mean_loss, mean_error_rate, current_step = self.end_batch(sess, False, rnn_state_reset_ratio=1.0)
        self.engine.zero_state()      # evaluation always resets the RNN state
        logging.info("Evaluation at step %d : loss %.5f - error_rate %.5f - duration %.2f",
                     current_step, mean_loss, mean_error_rate, time.time() - start_time)
        return mean_loss, mean_error_rate, current_step

    # ---- inference ---------------------------------------------------------------
    @_engine_stream
    def process_input(self, session, inputs, input_seq_lengths, run_options=None, run_metadata=None):
        """inputs [T_max, B, D], lengths [B] -> dense int prediction matrix padded with
        num_labels (:705-721); decoded by `self.decoder` (default: beam search, width 100, as the reference)."""
        x, dlen, _ = self._to_device(inputs, input_seq_lengths, np.zeros((self.batch_size, 1), np.int32))
        if self._async_beam is not None:
            self._async_beam.guard(torch.cuda.current_stream(self.engine.device))
        self.engine.forward(x, dlen, max_len=self._host_max(input_seq_lengths))
        pred = self._decode(dlen)          # (reads the result back: the stream is drained)
        self.engine.check()                # a bounded-wait time-out of the dataflow kernels invalidates the logits
        return pred

    def _decode(self, dlen):
        """Dense int32 prediction matrix [B, width] padded with num_labels."""
        if self.decoder == "beam":
            ids, out_len, _ = ops.ctc_beam_search(self.engine.logits, dlen, self.beam_width, self.merge_repeated)
        else:
            ids, out_len = ops.ctc_greedy_decode(self.engine.logits, dlen, ws=self.engine.ctc_ws)
            ids, out_len = ids.cpu().numpy(), out_len.cpu().numpy()
            if self.merge_repeated:
                ids, out_len = _merge_repeated(ids, out_len, self.num_labels)
        width = max(int(out_len.max()), 1)
        return ids[:, :width]

    def evaluate_full(self, sess, eval_dataset, input_seq_length, signal_processing, char_map,
                      run_options=None, run_metadata=None, n_mfcc=20, sample_rate=22050):
        audio = AudioProcessor(input_seq_length, signal_processing, n_mfcc=n_mfcc, load_sr=sample_rate)
        wer_list, cer_list = [], []
        feats, lens, texts = [], [], []
        B, T, D = self.batch_size, self.max_input_seq_length, audio.feature_size
        for n, (file, label, _) in enumerate(eval_dataset, 1):
            feat, length = audio.process_audio_file(file) if isinstance(file, str) else audio.process_signal(*file)
            if len(label) > self.max_target_seq_length or length > T:
                logging.warning("Warning - sample too long : %s (input : %d / text : %s)", file, length, len(label))
            else:
                padded = np.zeros((T, D), np.float32)
                padded[:len(feat)] = feat
                feats.append(padded); lens.append(length); texts.append(label)
            if n == len(eval_dataset):
                while len(feats) < B:
                    feats.append(np.zeros((T, D), np.float32)); lens.append(0); texts.append("")
            if len(feats) == B:
                pred = self.process_input(sess, np.swapaxes(np.stack(feats), 0, 1), lens)
                for row, truth in zip(pred, texts):
                    if len(truth) > 0:
                        hyp = _labels.get_labels_str(char_map, row)
                        wer_list.append(self.calculate_wer(hyp, truth) / float(len(truth.split())))
                        cer_list.append(self.calculate_cer(hyp, truth) / float(len(truth.replace(" ", ""))))
                feats, lens, texts = [], [], []
        wer = sum(wer_list) * 100 / float(len(wer_list))
        cer = sum(cer_list) * 100 / float(len(cer_list))
        return wer, cer


def _merge_repeated(ids, out_len, pad):
    """Collapse consecutive duplicate labels of each decoded row (TensorFlow's merge_repeated=True)."""
    out = np.full_like(ids, pad)
    lens = np.zeros_like(out_len)
    for b in range(ids.shape[0]):
        row = ids[b, :out_len[b]]
        if len(row):
            keep = np.concatenate(([True], row[1:] != row[:-1]))
            kept = row[keep]
            out[b, :len(kept)] = kept
            lens[b] = len(kept)
    return out, lens


def _edit_distance_tokens(r, h):
    """Levenshtein distance over arbitrary tokens (words / characters)."""
    vocab = {}
    a = [vocab.setdefault(t, len(vocab)) for t in r]
    b = [vocab.setdefault(t, len(vocab)) for t in h]
    return _edit_distance(a, b)
