"""GPU parity of the HIP audio front end (through the C ABI) against the numpy oracle and the
golden fbank fixtures generated from the reference's own numpy body."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import frontend as ofe  # noqa: E402  (checker only)


def synth(seed, n, sr):
    rng = np.random.RandomState(seed)
    t = np.arange(n) / float(sr)
    sig = 0.1 * rng.randn(n)
    for f0, a in ((220.0, 0.3), (1330.0, 0.2), (3100.0, 0.1)):
        sig += a * np.sin(2 * np.pi * f0 * (1 + 0.1 * seed) * t)
    return sig.astype(np.float32)


@pytest.mark.parametrize("sr,n_mfcc", [(16000, 20), (16000, 40), (22050, 20), (8000, 13)])
def test_mfcc_matches_oracle(sr, n_mfcc):
    from rnn_speech_amd.audioprocessor import AudioProcessor
    ap = AudioProcessor(10 ** 6, "mfcc", n_mfcc=n_mfcc)
    sigs = [synth(1, sr + 321, sr), synth(2, sr // 2 + 17, sr), synth(3, 2 * sr, sr)]   # ragged batch
    feat, lengths = ap.process_batch(sigs, sr, t_max=250)
    feat = feat.cpu().numpy()
    for b, sig in enumerate(sigs):
        ref = ofe.mfcc(sig, sr, n_mfcc=n_mfcc)
        assert lengths[b] == len(ref)
        n = min(len(ref), 250)
        # MFCCs are dB-scale numbers of magnitude 10..500; f32 DFT + log: 2e-3 absolute
        assert np.abs(feat[:n, b] - ref[:n]).max() < 2e-3, (b, np.abs(feat[:n, b] - ref[:n]).max())
        assert not feat[n:, b].any()                       # zero padding past the utterance


@pytest.mark.parametrize("tag", ["16k", "22k", "8k"])
def test_fbank_matches_reference_golden(golden_dir, tag):
    from rnn_speech_amd.audioprocessor import AudioProcessor
    z = np.load(os.path.join(golden_dir, "fbank_%s.npz" % tag))
    ap = AudioProcessor(10 ** 6, "fbank")
    feat, length = ap.process_signal(z["sig"], int(z["sr"]))
    assert length == int(z["length"]) and feat.shape == z["feat"].shape
    # static log-mel dims are pinned by the reference's numpy body; deltas by scipy savgol semantics
    assert np.abs(feat[:, :40] - z["feat"][:, :40]).max() < 2e-3
    assert np.abs(feat[:, 40:] - z["feat"][:, 40:]).max() < 2e-3


def test_truncation_contract(golden_dir):
    """Features are cut to max_input_seq_length, the returned length is NOT (util/audioprocessor.py:157-161)."""
    from rnn_speech_amd.audioprocessor import AudioProcessor
    z = np.load(os.path.join(golden_dir, "fbank_trunc.npz"))
    ap = AudioProcessor(int(z["max_len"]), "fbank")
    feat, length = ap.process_signal(z["sig"], int(z["sr"]))
    assert length == int(z["length"]) and length > int(z["max_len"])
    assert feat.shape == z["feat"].shape == (50, 120)
    # global statistics (mean, deltas at the far edge) still come from the WHOLE utterance
    assert np.abs(feat - z["feat"]).max() < 2e-3


def test_full_size_properties():
    """BASELINE size (32 x 10 s @ 16 kHz): frame count, linearity of the power path via a gain change
    (MFCC c0 shifts by 10*log10(g^2)*sqrt(128), other coefficients unchanged)."""
    from rnn_speech_amd.audioprocessor import AudioProcessor
    sr, n = 16000, 160000
    ap = AudioProcessor(1001, "mfcc", n_mfcc=40)
    sigs = [synth(b, n, sr) for b in range(4)]
    f1, l1 = ap.process_batch(sigs, sr)
    f2, l2 = ap.process_batch([4.0 * s for s in sigs], sr)
    assert l1 == [1001] * 4 == l2 and f1.shape == (1001, 4, 40)
    d = (f2 - f1).cpu().numpy()
    shift = 10 * np.log10(16.0) * np.sqrt(128.0)
    assert np.abs(d[..., 0] - shift).max() < 5e-2 and np.abs(d[..., 1:]).max() < 5e-2


def test_mfcc_front_end_at_headline_batch_matches_oracle():
    """BASELINE configs[1]'s front end exactly as bench.py runs it: B = 32 utterances of 10 s at 16 kHz, n_mfcc = 40 ->
    [1001, 32, 40]; every 5th row and two ragged ones against the oracle's librosa.feature.mfcc restatement
    (util/audioprocessor.py:63-75; the per-utterance top_db reference is a whole-utterance maximum, so a full-length check
    exercises the cross-workgroup atomicMax the short cases barely touch)."""
    from rnn_speech_amd.audioprocessor import AudioProcessor
    sr, n, B, T = 16000, 160000, 32, 1001
    ap = AudioProcessor(T, "mfcc", n_mfcc=40)
    sigs = [synth(b % 7, n, sr) * (0.2 + 0.1 * (b % 5)) for b in range(B)]
    sigs[4] = sigs[4][:100003]
    sigs[31] = sigs[31][:33333]
    sigs[9] = sigs[9].copy()
    sigs[9][60000:] *= 1e-4                                 # a loud start and a near-silent tail: the 80 dB floor binds
    feat, lengths = ap.process_batch(sigs, sr)
    assert feat.shape == (T, B, 40)
    feat = feat.cpu().numpy()
    for b in list(range(0, B, 5)) + [4, 9, 31]:
        ref = ofe.mfcc(sigs[b], sr, n_mfcc=40)
        assert lengths[b] == len(ref) == 1 + len(sigs[b]) // 160
        nb = min(len(ref), T)
        assert np.abs(feat[:nb, b] - ref[:nb]).max() < 3e-3, (b, np.abs(feat[:nb, b] - ref[:nb]).max())
        assert not feat[nb:, b].any()


# ------------------------------------------------------------------------ resampler + file path
@pytest.mark.parametrize("sr_in,sr_out", [(16000, 22050), (44100, 22050), (8000, 22050), (22050, 16000)])
def test_resampler_matches_oracle_and_scipy(sr_in, sr_out):
    """GPU kaiser_best sinc resampler vs the numpy restatement of resampy's algorithm (oracle), and -- as an
    independent sanity check on a band-limited signal -- vs scipy's polyphase resampler."""
    from math import gcd
    from scipy.signal import resample_poly
    from rnn_speech_amd import ops
    rng = np.random.RandomState(sr_in)
    n = [sr_in // 2, sr_in // 3 + 17]
    t = np.arange(max(n)) / float(sr_in)
    sigs = [(0.5 * np.sin(2 * np.pi * 440 * t[:k]) + 0.3 * np.sin(2 * np.pi * 1234.5 * t[:k] + 0.7)
             + 0.01 * rng.randn(k)).astype(np.float32) for k in n]
    host = np.zeros((2, max(n)), np.float32)
    for i, s in enumerate(sigs):
        host[i, :len(s)] = s
    out, n_out = ops.resample(torch.from_numpy(host).cuda(), n, sr_in, sr_out)
    out = out.cpu().numpy()
    for i, s in enumerate(sigs):
        ref = ofe.resample_kaiser_best(s, sr_in, sr_out)
        assert n_out[i] == len(ref) == int(np.ceil(len(s) * sr_out / sr_in))
        assert np.abs(out[i, :n_out[i]] - ref).max() < 2e-5
        assert not out[i, n_out[i]:].any()
        g = gcd(sr_in, sr_out)
        poly = resample_poly(s.astype(np.float64), sr_out // g, sr_in // g)
        m = min(len(poly), n_out[i])
        core = slice(400, m - 400)                                    # away from the edge transients
        assert np.abs(out[i, :m][core] - poly[:m][core]).max() < 0.02


def test_files_to_features_matches_signal_path(tmp_path):
    """process_files (decode -> GPU resample to 22,050 Hz -> front end; the reference's process_audio_file,
    util/audioprocessor.py:41-51) equals process_signal on the oracle-resampled waveform; mixed source rates
    and a padded short batch in one call."""
    import wave
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
    from flac_writer import write_flac
    from rnn_speech_amd.audioprocessor import AudioProcessor, DEFAULT_LOAD_SR
    rng = np.random.RandomState(5)
    files, sigs = [], []
    for i, (sr, kind) in enumerate([(16000, "wav"), (16000, "flac"), (22050, "wav")]):
        n = int(sr * (0.6 + 0.1 * i))
        t = np.arange(n) / float(sr)
        x = np.round((0.4 * np.sin(2 * np.pi * (300 + 100 * i) * t) + 0.05 * rng.randn(n)) * 20000).astype(np.int64)
        path = str(tmp_path / ("u%d.%s" % (i, kind)))
        if kind == "wav":
            with wave.open(path, "wb") as w:
                w.setnchannels(1)
                w.setsampwidth(2)
                w.setframerate(sr)
                w.writeframes(x.astype("<i2").tobytes())
        else:
            write_flac(path, x, sr, 16, blocksize=4096, plan=[{"kind": "fixed2", "porder": 3}])
        files.append(path)
        sigs.append((x.astype(np.float32) / np.float32(32768), sr))
    ap = AudioProcessor(200, "mfcc", n_mfcc=40)
    feat, lengths = ap.process_files(files, rows=4)
    assert feat.shape == (200, 4, 40) and lengths[3] == 0 and not feat[:, 3].any()
    for i, (s, sr) in enumerate(sigs):
        ref_sig = s if sr == DEFAULT_LOAD_SR else ofe.resample_kaiser_best(s, sr, DEFAULT_LOAD_SR).astype(np.float32)
        ref_feat, ref_len = ap.process_signal(ref_sig, DEFAULT_LOAD_SR)
        assert lengths[i] == ref_len
        got = feat[:ref_len, i].cpu().numpy()
        assert np.abs(got - ref_feat).max() < 2e-2 * max(1.0, np.abs(ref_feat).max())
    one, n1 = ap.process_audio_file(files[1])
    assert n1 == lengths[1] and np.abs(one - feat[:n1, 1].cpu().numpy()).max() < 1e-4


def test_sample_rates_above_32_khz_run_the_vector_alu_frame_kernel_and_match_the_oracle():
    """44.1 / 48 kHz `process_signal`: n_fft = round(0.025 sr) = 1102 / 1200 needs more LDS than a CU has for the matrix-core frame
    kernel, so these rates are what still reaches `frontend_frames_kernel` (the round-1 vector-ALU kernel) by default -- no GPU test
    did any more.  Both feature types, against the oracle."""
    from rnn_speech_amd.audioprocessor import AudioProcessor
    for sr in (44100, 48000):
        sig = synth(5, sr + 777, sr)
        ap = AudioProcessor(10 ** 6, "mfcc", n_mfcc=20)
        feat, length = ap.process_signal(sig, sr)
        ref = ofe.mfcc(sig, sr, n_mfcc=20)
        assert length == len(ref) and feat.shape == ref.shape
        assert np.abs(feat - ref).max() < 2e-3, (sr, np.abs(feat - ref).max())
    sig = synth(6, 44100 + 123, 44100)
    feat, length = AudioProcessor(10 ** 6, "fbank").process_signal(sig, 44100)
    ref = ofe.fbank(sig, 44100)
    assert length == len(ref) and np.abs(feat - ref).max() < 2e-3


def test_vector_alu_front_end_switch_keeps_parity():
    """AMDSPEECH_FRONTEND_MFMA=0 (run-time switches): the whole front end on the round-1 kernels, in a child process
    (the library reads the switch once): the oracle / golden cases of this file again."""
    import subprocess
    import sys
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-m", "gpu", "-q", "-x", "-k",
                          "mfcc_matches_oracle or fbank_matches_reference_golden or truncation_contract"],
                         env=dict(os.environ, AMDSPEECH_FRONTEND_MFMA="0"), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
