"""AudioProcessor drop-in (reference: /root/reference/util/audioprocessor.py:11-61).

Same constructor, attributes and return contract -- (features [T', D] truncated to
max_input_seq_length, UNtruncated frame count) -- but the features are computed by the
HIP front-end kernels (csrc/frontend.hip) instead of librosa/numpy.  `process_batch`
is the fast path: a whole mini-batch of signals -> one time-major device tensor.
"""
import numpy as np
import torch

from . import ops

FRAME_STRIDE = 0.01
FRAME_SIZE = 0.025
DEFAULT_LOAD_SR = 22050   # librosa.load default used by the reference's file path (:49)


class AudioProcessor(object):
    def __init__(self, max_input_seq_length, feature_type="mfcc", n_mfcc=20, device="cuda", load_sr=DEFAULT_LOAD_SR):
        """feature_type: 'mfcc' (n_mfcc-dim, reference default 20) or 'fbank' (120-dim).  load_sr: the rate audio
        FILES are resampled to before feature extraction (config.ini `sample_rate`; the reference's librosa.load
        default, 22,050 Hz) -- set it to the corpus rate (16,000 for LibriSpeech) to skip resampling."""
        self.max_input_seq_length = max_input_seq_length
        self.load_sr = int(load_sr)
        self.feature_type = feature_type
        self.device = device
        if feature_type == "mfcc":
            self.feature_size = int(n_mfcc)
        elif feature_type == "fbank":
            self.feature_size = 120
        else:
            raise ValueError("{0} is not a valid extraction function, only fbank and mfcc are accepted."
                             .format(feature_type))
        self.n_mfcc = int(n_mfcc)

    @staticmethod
    def get_mfcc_length_from_duration(duration):
        """Estimate only (reference :30-39)."""
        return int(duration // FRAME_STRIDE) - 1

    # ---- reference surface ----------------------------------------------------
    def process_audio_file(self, file_name):
        feat, lengths = self.process_files([file_name])
        n = min(lengths[0], self.max_input_seq_length)
        return feat[:n, 0, :].cpu().numpy(), lengths[0]

    def process_signal(self, sig, sr):
        feat, lengths = self.process_batch([np.asarray(sig, dtype=np.float32)], sr)
        n = min(lengths[0], self.max_input_seq_length)
        return feat[:n, 0, :].cpu().numpy(), lengths[0]

    # ---- batched device path ----------------------------------------------------
    def stage(self, signals, rows=None):
        """Host half of an upload, safe to run on a producer thread: the signals packed into one [rows, width] float32 block in
        PINNED memory (a small pool of blocks, each reused once the copy that read it has completed), so that the device half is
        one asynchronous DMA instead of a pageable copy the host thread has to sit through (20 MB: 5 ms)."""
        n = [len(s) for s in signals]
        shape = (rows or len(signals), max(max(n) if n else 0, 1))
        block = _PINNED.get(shape) if (self.device != "cpu" and torch.cuda.is_available()) else _Staged(torch.zeros(shape))
        host = block.tensor.numpy()
        for i, s in enumerate(signals):
            host[i, :len(s)] = s
            host[i, len(s):] = 0.0
        host[len(signals):] = 0.0
        return block, n

    def _upload_staged(self, block):
        dev = block.tensor.to(self.device, non_blocking=True)
        block.mark_in_flight()
        return dev

    def _upload(self, signals, rows=None):
        block, n = self.stage(signals, rows)
        return self._upload_staged(block), n

    def process_batch(self, signals, sr, t_max=None, staged=None):
        """signals: list of 1-D float arrays, all at sample rate `sr`.  Returns (feat [t_max, B, D] device
        float32, zero past each utterance; list of UNtruncated frame counts).  staged: (block, n) from stage()."""
        t_max = self.max_input_seq_length if t_max is None else t_max
        if staged is not None:
            pcm, n = self._upload_staged(staged[0]), staged[1]
        else:
            pcm, n = self._upload(signals)
        return ops.frontend(pcm, n, int(sr), self.feature_type, int(t_max), self.n_mfcc)

    def stage_files(self, decoded):
        """stage() for decoded files, grouped by source rate as process_files uploads them: [(sr, idx, block, n)]."""
        by_rate = {}
        for i, (_, sr) in enumerate(decoded):
            by_rate.setdefault(int(sr), []).append(i)
        return [(sr, idx) + self.stage([decoded[i][0] for i in idx]) for sr, idx in by_rate.items()]

    def process_files(self, file_names, t_max=None, rows=None, decoded=None, staged=None):
        """What the reference's dataset map does per file (process_audio_file: librosa.load at 22,050 Hz,
        then the extractor, util/audioprocessor.py:41-61), for a whole mini-batch: files are decoded natively
        on host threads (or passed in as `decoded` [(signal, sr), ...]), uploaded once, resampled to
        22,050 Hz on the GPU per source rate and handed to the front-end kernels without leaving HBM.
        `rows` > len(files) pads the batch with empty utterances (length 0)."""
        t_max = self.max_input_seq_length if t_max is None else t_max
        if decoded is None:
            decoded = decode_files(file_names)
        B = rows or len(decoded)
        if staged is None:
            staged = self.stage_files(decoded)
        parts, lengths = [], [0] * B
        for sr, idx, block, n in staged:
            pcm = self._upload_staged(block)
            if sr != self.load_sr:
                pcm, n = ops.resample(pcm, n, sr, self.load_sr)
            parts.append((idx, pcm, n))
        width = max(max(p[1].shape[1] for p in parts), 1) if parts else 1
        if len(parts) == 1 and len(parts[0][0]) == B:
            pcm, n = parts[0][1], parts[0][2]
        else:
            pcm = torch.zeros(B, width, device=self.device)
            n = [0] * B
            for idx, part, lens in parts:
                ii = torch.as_tensor(idx, device=self.device)
                pcm[ii, :part.shape[1]] = part
                for j, i in enumerate(idx):
                    n[i] = lens[j]
        # (an empty row has no frames; the front end wants > n_fft/2 samples for real ones)
        return ops.frontend(pcm, n, self.load_sr, self.feature_type, int(t_max), self.n_mfcc)


class _Staged(object):
    """A [rows, width] host block handed out by the staging pool.  Claimed from get() until mark_in_flight() records the event
    behind the copy that reads it; free again once that event has completed."""

    def __init__(self, tensor, flat=None):
        self.tensor, self.flat = tensor, flat
        self.event, self.claimed = None, True

    def mark_in_flight(self):
        if self.tensor.is_pinned():
            self.event = torch.cuda.Event()
            self.event.record(torch.cuda.current_stream())
        self.claimed = False

    def free(self):
        return not self.claimed and (self.event is None or self.event.query())


class _PinnedPool(object):
    """Pinned staging blocks, reused: a hipHostMalloc per mini-batch would cost more than the copy it speeds up."""

    def __init__(self, limit=12):
        import threading
        self._lock = threading.Lock()
        self._blocks, self._limit = [], limit

    def get(self, shape):
        need = shape[0] * shape[1]
        with self._lock:
            for i, blk in enumerate(self._blocks):
                if blk.flat.numel() >= need and blk.free():
                    self._blocks[i] = _Staged(blk.flat[:need].view(shape), blk.flat)
                    return self._blocks[i]
            if len(self._blocks) >= self._limit:      # make room: forget a free block (too small, or just the oldest)
                for i, blk in enumerate(self._blocks):
                    if blk.free():
                        del self._blocks[i]
                        break
        flat = torch.empty(need, dtype=torch.float32).pin_memory()
        blk = _Staged(flat.view(shape), flat)
        with self._lock:
            if len(self._blocks) < self._limit:
                self._blocks.append(blk)
        return blk


_PINNED = _PinnedPool()
_POOL = None


def decode_files(file_names):
    """[(mono float32 signal, sample_rate), ...] -- native decoders (WAVE / FLAC / NIST SPHERE) on a
    thread pool; the C call releases the GIL."""
    global _POOL
    if len(file_names) <= 1:
        return [ops.audio_decode(f) for f in file_names]
    if _POOL is None:
        import os
        from concurrent.futures import ThreadPoolExecutor
        _POOL = ThreadPoolExecutor(max_workers=min(32, os.cpu_count() or 1))
    return list(_POOL.map(ops.audio_decode, file_names))


def load_audio(file_name, target_sr=DEFAULT_LOAD_SR, device="cuda"):
    """librosa.load(file, sr=target_sr) equivalent: mono float32 numpy at target_sr (native decode, GPU
    resampler with resampy kaiser_best semantics; not bit-compatible with any particular librosa release)."""
    sig, sr = ops.audio_decode(file_name)
    if sr != target_sr:
        pcm = torch.from_numpy(np.ascontiguousarray(sig[None, :])).to(device)
        out, n = ops.resample(pcm, [len(sig)], sr, target_sr)
        sig = out[0, :n[0]].cpu().numpy()
    return sig, target_sr
