"""Native audio decoders (csrc/audio_io.cpp; SURVEY 8f-3: the decode half of librosa.load at
util/audioprocessor.py:49).  Host code: runs without a GPU.  WAV is checked against the `wave` module /
numpy, SPHERE against hand-built files, FLAC against streams from tests/flac_writer.py (an encoder
written from the specification) with the STREAMINFO MD5 (hashlib) as the end-to-end check."""
import wave

import numpy as np
import pytest

from corpus_fixture import write_sphere
from flac_writer import write_flac


@pytest.fixture(scope="module")
def ops():
    from rnn_speech_amd import ops as _ops
    return _ops


def _signal(n, nch, bps, seed=0, kind="speechlike"):
    rng = np.random.RandomState(seed)
    t = np.arange(n)
    amp = (1 << (bps - 1)) - 1
    if kind == "noise":
        x = rng.randint(-amp, amp, size=(n, nch))
    else:
        base = 0.4 * np.sin(2 * np.pi * t / 53.0) + 0.2 * np.sin(2 * np.pi * t / 7.3 + 1.0) + 0.02 * rng.randn(n)
        x = np.stack([(base * (0.9 - 0.3 * c) + 0.01 * rng.randn(n)) * amp for c in range(nch)], axis=1)
    return np.clip(np.round(x), -amp - 1, amp).astype(np.int64)


def _expect(x, bps):
    return (x.astype(np.float32) / np.float32(1 << (bps - 1))).mean(axis=1, dtype=np.float32) if x.shape[1] > 1 \
        else x[:, 0].astype(np.float32) / np.float32(1 << (bps - 1))


@pytest.mark.parametrize("nch,bps,plan", [
    (1, 16, [{"kind": "fixed2", "porder": 3}]),
    (1, 16, [{"kind": "fixed0"}, {"kind": "fixed1", "porder": 2}, {"kind": "fixed3", "porder": 4}, {"kind": "fixed4"}]),
    (1, 16, [{"kind": "lpc", "porder": 2, "lpc_order": 8}, {"kind": "lpc", "lpc_order": 12, "porder": 3},
             {"kind": "lpc", "lpc_order": 1}, {"kind": "lpc", "lpc_order": 32, "porder": 1}]),
    (1, 16, [{"kind": "verbatim"}, {"kind": "fixed2", "escape": True, "porder": 2}]),
    (2, 16, [{"kind": "fixed2", "stereo": "left_side", "porder": 2}, {"kind": "lpc", "stereo": "side_right"},
             {"kind": "fixed1", "stereo": "mid_side", "porder": 1}, {"kind": "fixed2", "stereo": "indep"}]),
    (2, 24, [{"kind": "lpc", "stereo": "mid_side", "porder": 3}]),
    (1, 8, [{"kind": "fixed1"}]),
    (1, 16, [{"kind": "fixed2", "rate_in_frame": False, "bits_in_frame": False}]),
])
def test_flac_decoder_roundtrip_with_md5(ops, tmp_path, nch, bps, plan):
    x = _signal(3000 + 137, nch, bps, seed=nch * 10 + bps)          # last block is short and odd-sized
    path = str(tmp_path / "t.flac")
    write_flac(path, x, 16000, bps, blocksize=1024, plan=plan)
    assert ops.audio_probe(path) == (16000, nch, len(x))
    y, sr = ops.audio_decode(path, verify=True)                     # frame CRCs + MD5 of the decoded samples
    assert sr == 16000 and y.dtype == np.float32 and len(y) == len(x)
    assert np.array_equal(y, _expect(x, bps))


def test_flac_special_blocks_and_corruption(ops, tmp_path):
    n = 4096
    x = np.zeros((n, 1), np.int64)
    x[1024:2048, 0] = 1234                                           # a CONSTANT block
    x[2048:3072, 0] = _signal(1024, 1, 16, 3)[:, 0] & ~0xF            # 4 wasted bits
    x[3072:, 0] = _signal(1024, 1, 16, 4, "noise")[:, 0]             # incompressible: Rice parameters near bps
    path = str(tmp_path / "s.flac")
    write_flac(path, x, 22050, 16, blocksize=1024, plan=[{"kind": "auto"}, {"kind": "constant"},
                                                           {"kind": "fixed2", "porder": 2}, {"kind": "fixed0", "porder": 3}])
    y, sr = ops.audio_decode(path, verify=True)
    assert sr == 22050 and np.array_equal(y, _expect(x, 16))
    raw = bytearray(open(path, "rb").read())
    bad = str(tmp_path / "bad.flac")
    raw[len(raw) // 2] ^= 0x10                                        # flip one bit inside a frame
    open(bad, "wb").write(bytes(raw))
    from rnn_speech_amd.lib import AmdSpeechError
    with pytest.raises(AmdSpeechError, match="flac"):
        ops.audio_decode(bad)
    # a wrong signature is caught only when verification is requested
    raw = bytearray(open(path, "rb").read())
    raw[4 + 4 + 20] ^= 0xFF
    open(bad, "wb").write(bytes(raw))
    ops.audio_decode(bad)
    with pytest.raises(AmdSpeechError, match="MD5"):
        ops.audio_decode(bad, verify=True)


@pytest.mark.parametrize("width,nch", [(1, 1), (2, 1), (2, 2), (3, 2), (4, 1)])
def test_wav_pcm_matches_wave_module(ops, tmp_path, width, nch):
    n = 2000
    bps = 8 * width
    x = _signal(n, nch, bps, seed=width)
    path = str(tmp_path / "t.wav")
    with wave.open(path, "wb") as w:
        w.setnchannels(nch)
        w.setsampwidth(width)
        w.setframerate(8000)
        if width == 1:
            w.writeframes((x + 128).astype(np.uint8).tobytes())
        elif width == 3:
            w.writeframes(b"".join(int(v).to_bytes(3, "little", signed=True) for v in x.reshape(-1)))
        else:
            w.writeframes(x.astype("<i%d" % width).tobytes())
    assert ops.audio_probe(path) == (8000, nch, n)
    y, sr = ops.audio_decode(path)
    assert sr == 8000 and np.array_equal(y, _expect(x, bps))


def test_wav_float_extensible_and_odd_chunks(ops, tmp_path):
    import struct
    x = (np.random.RandomState(0).rand(500, 2).astype(np.float32) - 0.5)
    body = x.astype("<f4").tobytes()
    fmt = struct.pack("<HHIIHHHHIH14s", 0xFFFE, 2, 44100, 44100 * 8, 8, 32, 22, 32, 3, 3,
                      b"\x00\x00\x00\x00\x10\x00\x80\x00\x00\xaa\x00\x38\x9b\x71")
    chunks = b"LIST" + struct.pack("<I", 3) + b"abc\x00" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + \
        b"data" + struct.pack("<I", len(body)) + body
    path = str(tmp_path / "f.wav")
    open(path, "wb").write(b"RIFF" + struct.pack("<I", 4 + len(chunks)) + b"WAVE" + chunks)
    y, sr = ops.audio_decode(path)
    assert sr == 44100 and np.array_equal(y, (x[:, 0] + x[:, 1]) / np.float32(2))


def test_sphere_both_byte_orders_and_errors(ops, tmp_path):
    x = _signal(1234, 1, 16, 9)[:, 0].astype(np.int16)
    for order in ("01", "10"):
        path = str(tmp_path / ("t%s.sph" % order))
        write_sphere(path, x, rate=16000, byte_format=order)
        assert ops.audio_probe(path) == (16000, 1, len(x))
        y, sr = ops.audio_decode(path)
        assert sr == 16000 and np.array_equal(y, x.astype(np.float32) / np.float32(32768))
    from rnn_speech_amd.lib import AmdSpeechError
    junk = str(tmp_path / "x.mp3")
    open(junk, "wb").write(b"ID3\x03" + bytes(64))
    with pytest.raises(AmdSpeechError):
        ops.audio_decode(junk)
    with pytest.raises(AmdSpeechError):
        ops.audio_decode(str(tmp_path / "missing.wav"))


def test_malformed_files_never_abort_the_process(ops, tmp_path):
    """Byte-mutation fuzz of valid FLAC / WAVE / SPHERE files (run in a child process: before the C ABI caught
    C++ exceptions a STREAMINFO with a huge total_samples made `reserve` throw bad_alloc through extern "C" and
    std::terminate killed the interpreter).  Every mutant must either decode or raise AmdSpeechError."""
    import subprocess
    import sys
    import os
    x = _signal(2000, 1, 16, seed=3)
    flac = str(tmp_path / "f.flac")
    write_flac(flac, x, 16000, 16, blocksize=512, plan=[{"kind": "lpc", "lpc_order": 8, "porder": 2}])
    wav = str(tmp_path / "f.wav")
    with wave.open(wav, "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000); w.writeframes(x.astype("<i2").tobytes())
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = r"""
import sys, numpy as np
sys.path.insert(0, %r)
from rnn_speech_amd import ops
from rnn_speech_amd.lib import AmdSpeechError
ok = bad = 0
for src in (%r, %r):
    raw = bytearray(open(src, "rb").read())
    rng = np.random.RandomState(len(raw))
    for trial in range(400):
        m = bytearray(raw)
        if trial == 0 and src.endswith(".flac"):
            m[8 + 13:8 + 18] = b"\xff\xff\xff\xff\xff"[:5]      # STREAMINFO total_samples = 2**36 - 1 (and 32 bps bits)
        else:
            for _ in range(rng.randint(1, 4)):
                m[rng.randint(0, min(len(m), 200 if trial %% 2 else len(m)))] = rng.randint(0, 256)
        path = src + ".mut"
        open(path, "wb").write(bytes(m))
        try:
            ops.audio_decode(path, verify=bool(trial & 1))
            ok += 1
        except AmdSpeechError:
            bad += 1
print("survived", ok, bad)
""" % (root, flac, wav)
    out = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=280)
    assert out.returncode == 0 and "survived" in out.stdout, out.stdout[-1500:] + out.stderr[-1500:]
    bad = int(out.stdout.split()[-1])
    assert bad > 50                                                  # the mutations do hit the parsers


def test_crc_tables_are_thread_safe(ops):
    """amdspeech_crc32c and the FLAC CRC-16 build their tables on first use; first use from many threads at once
    (the decode pool) must give the known answers."""
    import ctypes
    import subprocess
    import sys
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = r"""
import sys, ctypes, threading
sys.path.insert(0, %r)
from rnn_speech_amd import lib
l = lib.load()
data = b"123456789"
res = []
def work():
    res.append(l.amdspeech_crc32c(ctypes.c_char_p(data), len(data), 0))
ths = [threading.Thread(target=work) for _ in range(32)]
[t.start() for t in ths]; [t.join() for t in ths]
assert all(r == 0xE3069283 for r in res), res          # CRC-32C check value
print("ok")
""" % root
    out = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "ok" in out.stdout, out.stdout + out.stderr
