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
