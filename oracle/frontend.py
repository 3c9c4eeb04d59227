"""ORACLE (test infrastructure only) -- audio front-end restatement (numpy, float64).

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may
import this package; the product path never does.

Follows:
  * mfcc mode ... /root/reference/util/audioprocessor.py:63-75, which calls
    `librosa.feature.mfcc(sig, sr, hop_length=round(sr*0.01), n_fft=round(sr*0.025))`.
    librosa is an un-vendored, un-pinned dependency (requirements.txt:1) that is
    NOT importable here, so its published algorithm is restated:
    centred reflect-padded STFT with a periodic Hann window -> |.|^2 ->
    128 Slaney mel filters (area-normalised) -> 10*log10(max(.,1e-10)) clamped to
    (utterance max - 80 dB) -> orthonormal DCT-II -> first n_mfcc rows.
    PARITY UNPINNED at the librosa boundary (no golden vectors exist in the
    reference); cross-checked piecewise against scipy.fft / scipy.signal and, as
    a whole chain, against transformers.audio_utils (an independent
    librosa-compatible implementation present in the image:
    tests/test_cpu_oracle.py::test_mfcc_against_transformers_audio_utils).
  * fbank mode .. /root/reference/util/audioprocessor.py:77-161.  The static
    40 log-mel dims are the reference's own numpy code and ARE pinned by
    fixtures generated from the imported reference (tests/golden/fbank_*.npz,
    tools/make_golden.py).  The delta / delta-delta dims call
    `librosa.feature.delta` (:148-149), restated with librosa>=0.6 semantics
    (Savitzky-Golay, width 9, polyorder 1, deriv 1, mode 'interp'):
    PARITY UNPINNED for those 80 dims.
"""
import numpy as np

FRAME_STRIDE = 0.01   # util/audioprocessor.py:6
FRAME_SIZE = 0.025    # util/audioprocessor.py:7


def hop_and_window(sr):
    # int(round()) is Python banker's rounding: 22050 Hz -> hop 220, window 551
    return int(round(sr * FRAME_STRIDE)), int(round(sr * FRAME_SIZE))


# ----------------------------------------------------------------------------
# librosa-semantics pieces (mfcc mode)
# ----------------------------------------------------------------------------
def _hz_to_mel_slaney(f):
    f = np.asarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    big = f >= min_log_hz
    mels = np.where(big, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, mels)
    return mels


def _mel_to_hz_slaney(m):
    m = np.asarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    freqs = f_sp * m
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    big = m >= min_log_mel
    return np.where(big, min_log_hz * np.exp(logstep * (m - min_log_mel)), freqs)


def slaney_mel_filterbank(sr, n_fft, n_mels=128):
    """librosa.filters.mel(sr, n_fft, n_mels, fmin=0, fmax=sr/2, htk=False, norm='slaney')."""
    n_bins = 1 + n_fft // 2
    fftfreqs = np.linspace(0.0, sr / 2.0, n_bins)
    mel_pts = np.linspace(_hz_to_mel_slaney(0.0), _hz_to_mel_slaney(sr / 2.0), n_mels + 2)
    mel_f = _mel_to_hz_slaney(mel_pts)
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    w = np.zeros((n_mels, n_bins))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0.0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    return w * enorm[:, None]


def periodic_hann(n):
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)


def dct2_ortho_matrix(n_out, n_in):
    """Rows k=0..n_out-1 of the orthonormal DCT-II over n_in points."""
    k = np.arange(n_out)[:, None]
    n = np.arange(n_in)[None, :]
    m = np.cos(np.pi * k * (2 * n + 1) / (2.0 * n_in)) * np.sqrt(2.0 / n_in)
    m[0] *= np.sqrt(0.5)
    return m


def power_spectrogram_centered(sig, n_fft, hop):
    """|STFT|^2, centre=True, reflect padding, periodic Hann; returns [T, n_fft//2+1]."""
    sig = np.asarray(sig, dtype=np.float64)
    pad = n_fft // 2
    y = np.pad(sig, (pad, pad), mode="reflect")
    n_frames = 1 + (len(y) - n_fft) // hop
    idx = np.arange(n_fft)[None, :] + hop * np.arange(n_frames)[:, None]
    frames = y[idx] * periodic_hann(n_fft)[None, :]
    spec = np.fft.rfft(frames, n=n_fft, axis=1)
    return spec.real ** 2 + spec.imag ** 2


def mfcc(sig, sr, n_mfcc=20, n_mels=128, top_db=80.0):
    """Returns [T, n_mfcc] float64; T = 1 + (N + 2*(n_fft//2) - n_fft)//hop."""
    hop, n_fft = hop_and_window(sr)
    power = power_spectrogram_centered(sig, n_fft, hop)            # [T, bins]
    mel = power @ slaney_mel_filterbank(sr, n_fft, n_mels).T       # [T, n_mels]
    db = 10.0 * np.log10(np.maximum(mel, 1e-10))
    db = np.maximum(db, db.max() - top_db)
    return db @ dct2_ortho_matrix(n_mfcc, n_mels).T


# ----------------------------------------------------------------------------
# fbank mode (the reference's own numpy body)
# ----------------------------------------------------------------------------
def htk_fbank_matrix(sr, nfft=512, nfilt=40):
    """util/audioprocessor.py:107-133 -- 40 triangles on the HTK mel scale."""
    high_mel = 2595.0 * np.log10(1.0 + (float(sr) / 2.0) / 700.0)
    mel_points = np.linspace(0.0, high_mel, nfilt + 2)
    hz_points = 700.0 * (10.0 ** (mel_points / 2595.0) - 1.0)
    edge = np.floor((nfft + 1) * hz_points / sr)
    fb = np.zeros((nfilt, nfft // 2 + 1))
    for m in range(1, nfilt + 1):
        lo, ce, hi = int(edge[m - 1]), int(edge[m]), int(edge[m + 1])
        for k in range(lo, ce):
            fb[m - 1, k] = (k - edge[m - 1]) / (edge[m] - edge[m - 1])
        for k in range(ce, hi):
            fb[m - 1, k] = (edge[m + 1] - k) / (edge[m + 1] - edge[m])
    return fb


def fbank_static(sig, sr, nfft=512, nfilt=40):
    """util/audioprocessor.py:87-147 -> [40, T] mean-normalised log-mel (float64)."""
    # :87 runs in the INPUT dtype (float32 audio stays float32 through the
    # pre-emphasis; only the zero-padding at :94-96 promotes to float64)
    sig = np.asarray(sig)
    emph = np.empty_like(sig)
    emph[0] = sig[0]
    emph[1:] = sig[1:] - sig.dtype.type(0.97) * sig[:-1]
    hop, win = hop_and_window(sr)
    n = len(emph)
    n_frames = int(np.ceil(abs(n - win) / float(hop)))
    padded = np.zeros(n_frames * hop + win)
    padded[:n] = emph
    idx = np.arange(win)[None, :] + hop * np.arange(n_frames)[:, None]
    frames = padded[idx] * np.hamming(win)[None, :]
    # np.fft.rfft(frames, 512): zero-pads when win < 512, TRUNCATES when win > 512
    mag = np.abs(np.fft.rfft(frames, nfft, axis=1))
    power = (1.0 / nfft) * mag ** 2
    fb = power @ htk_fbank_matrix(sr, nfft, nfilt).T
    fb = np.where(fb == 0, np.finfo(float).eps, fb)
    fb = 10.0 * np.log10(fb)
    fb = fb - (fb.mean(axis=0) + 1e-8)
    return fb.T


def delta_savgol9(x):
    """librosa.feature.delta (>=0.6) along the last axis: Savitzky-Golay width 9,
    polyorder 1, deriv 1, mode 'interp'.  Interior: sum_k k*x[t+k]/60; the first
    and last 4 frames take the slope of the line fitted to the first/last 9."""
    x = np.asarray(x, dtype=np.float64)
    t = x.shape[-1]
    if t < 9:
        raise ValueError("delta needs at least 9 frames (librosa raises too)")
    k = np.arange(-4, 5, dtype=np.float64)
    out = np.empty_like(x)
    for j in range(4, t - 4):
        out[..., j] = (x[..., j - 4:j + 5] * k).sum(-1) / 60.0
    out[..., :4] = ((x[..., :9] * k).sum(-1) / 60.0)[..., None]
    out[..., t - 4:] = ((x[..., t - 9:] * k).sum(-1) / 60.0)[..., None]
    return out


def fbank(sig, sr):
    """Returns [T, 120] float64: static | delta | delta-delta."""
    static = fbank_static(sig, sr)
    d1 = delta_savgol9(static)
    d2 = delta_savgol9(d1)
    return np.vstack([static, d1, d2]).T


def extract(sig, sr, feature_type="mfcc", max_input_seq_length=None, n_mfcc=20):
    """AudioProcessor.process_signal contract (util/audioprocessor.py:52-75,157-161):
    returns (features truncated to max_input_seq_length, UNtruncated length)."""
    feat = mfcc(sig, sr, n_mfcc=n_mfcc) if feature_type == "mfcc" else fbank(sig, sr)
    length = len(feat)
    if max_input_seq_length is not None and length > max_input_seq_length:
        feat = feat[:max_input_seq_length]
    return feat, length


# ----------------------------------------------------------------------------
# librosa.load's resampling step (util/audioprocessor.py:49: librosa.load(file) -> sr 22050)
# ----------------------------------------------------------------------------
def resample_kaiser_best(x, sr_orig, sr_new):
    """librosa.resample(x, sr_orig, sr_new, res_type='kaiser_best', fix=True, scale=False) as implemented by
    resampy (interpolated windowed-sinc table: 64 zero crossings, 2**9 entries per crossing, roll-off
    0.9475937167399596, Kaiser beta 14.769656459379492), restated from the published algorithm -- resampy /
    librosa are not importable here: PARITY UNPINNED.  float64 throughout."""
    x = np.asarray(x, np.float64)
    ratio = float(sr_new) / float(sr_orig)
    num_zeros, precision = 64, 9
    rolloff, beta = 0.9475937167399596, 14.769656459379492
    num_bits = 2 ** precision
    n = num_bits * num_zeros
    sinc_win = rolloff * np.sinc(rolloff * np.linspace(0, num_zeros, num=n + 1, endpoint=True))
    taper = np.kaiser(2 * n + 1, beta)[n:]
    win = taper * sinc_win
    if ratio < 1:
        win = win * ratio
    delta = np.zeros_like(win)
    delta[:-1] = np.diff(win)
    scale = min(1.0, ratio)
    step = int(scale * num_bits)
    n_out = int(len(x) * ratio)
    y = np.zeros(int(np.ceil(len(x) * ratio)))
    nwin = len(win)
    for t in range(n_out):
        tr = t / ratio
        k = int(tr)
        frac = scale * (tr - k)
        idx = frac * num_bits
        off = int(idx)
        eta = idx - off
        cnt = min(k + 1, (nwin - off) // step)
        j = off + step * np.arange(cnt)
        acc = np.dot(win[j] + eta * delta[j], x[k - np.arange(cnt)]) if cnt > 0 else 0.0
        frac = scale - frac
        idx = frac * num_bits
        off = int(idx)
        eta = idx - off
        cnt = min(len(x) - k - 1, (nwin - off) // step)
        if cnt > 0:
            j = off + step * np.arange(cnt)
            acc += np.dot(win[j] + eta * delta[j], x[k + 1 + np.arange(cnt)])
        y[t] = acc
    return y
