"""AudioProcessor drop-in (reference: /root/reference/util/audioprocessor.py:11-61).

Same constructor, attributes and return contract -- (features [T', D] truncated to
max_input_seq_length, UNtruncated frame count) -- but the features are computed by the
HIP front-end kernels (csrc/frontend.hip) instead of librosa/numpy.  `process_batch`
is the fast path: a whole mini-batch of signals -> one time-major device tensor.
"""
import wave

import numpy as np
import torch

from . import ops

FRAME_STRIDE = 0.01
FRAME_SIZE = 0.025
DEFAULT_LOAD_SR = 22050   # librosa.load default used by the reference's file path (:49)


class AudioProcessor(object):
    def __init__(self, max_input_seq_length, feature_type="mfcc", n_mfcc=20, device="cuda"):
        """feature_type: 'mfcc' (n_mfcc-dim, reference default 20) or 'fbank' (120-dim)."""
        self.max_input_seq_length = max_input_seq_length
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
        sig, sr = load_audio(file_name, DEFAULT_LOAD_SR)
        return self.process_signal(sig, sr)

    def process_signal(self, sig, sr):
        feat, lengths = self.process_batch([np.asarray(sig, dtype=np.float32)], sr)
        n = min(lengths[0], self.max_input_seq_length)
        return feat[:n, 0, :].cpu().numpy(), lengths[0]

    # ---- batched device path ----------------------------------------------------
    def process_batch(self, signals, sr, t_max=None):
        """signals: list of 1-D float arrays.  Returns (feat [t_max, B, D] device float32,
        zero past each utterance; list of UNtruncated frame counts)."""
        t_max = self.max_input_seq_length if t_max is None else t_max
        n = [len(s) for s in signals]
        n_max = max(max(n), 1)
        host = np.zeros((len(signals), n_max), np.float32)
        for i, s in enumerate(signals):
            host[i, :len(s)] = s
        pcm = torch.from_numpy(host).to(self.device)
        return ops.frontend(pcm, n, int(sr), self.feature_type, int(t_max), self.n_mfcc)


def load_audio(file_name, target_sr):
    """Mono float32 at target_sr.  WAV (PCM 8/16/32-bit) only; other containers and the
    reference's exact librosa resampler are a 'next' row.  The polyphase
    resampler here is NOT bit-compatible with librosa.load."""
    with wave.open(file_name, "rb") as w:
        sr, nch, width, nfr = w.getframerate(), w.getnchannels(), w.getsampwidth(), w.getnframes()
        raw = w.readframes(nfr)
    if width == 2:
        x = np.frombuffer(raw, "<i2").astype(np.float32) / 32768.0
    elif width == 4:
        x = np.frombuffer(raw, "<i4").astype(np.float32) / 2147483648.0
    elif width == 1:
        x = (np.frombuffer(raw, "u1").astype(np.float32) - 128.0) / 128.0
    else:
        raise ValueError("unsupported WAV sample width %d" % width)
    if nch > 1:
        x = x.reshape(-1, nch).mean(axis=1)
    if sr != target_sr:
        from math import gcd
        from scipy.signal import resample_poly
        g = gcd(int(sr), int(target_sr))
        x = resample_poly(x, target_sr // g, sr // g).astype(np.float32)
    return x, target_sr
