"""Dev tool: the FILE path of the input pipeline at full size: 32 FLAC + 32 WAV files of 10 s at 16 kHz -> native decode on host
threads -> upload -> GPU resampler (22,050 Hz, librosa.load semantics) -> front end; per-stage wall time."""
import os, sys, time, tempfile, wave
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from flac_writer import write_flac
from rnn_speech_amd.audioprocessor import AudioProcessor, decode_files
d = tempfile.mkdtemp()
rng = np.random.RandomState(0)
files = []
t0 = time.time()
for i in range(32):
    pcm = (rng.randn(160000) * 3000).astype(np.int16)
    p = os.path.join(d, "u%02d.flac" % i); write_flac(p, pcm.reshape(-1, 1), 16000, 16); files.append(p)
for i in range(32):
    pcm = (rng.randn(160000) * 3000).astype(np.int16)
    p = os.path.join(d, "u%02d.wav" % i)
    with wave.open(p, "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000); w.writeframes(pcm.tobytes())
    files.append(p)
print("wrote 64 files in %.1f s (python encoder)" % (time.time() - t0))
for name, fl in (("flac", files[:32]), ("wav", files[32:])):
    decode_files(fl)
    t0 = time.time(); dec = decode_files(fl); t_dec = time.time() - t0
    ap = AudioProcessor(1400, "mfcc", n_mfcc=40)
    ap.process_files(None, decoded=dec); torch.cuda.synchronize()
    t0 = time.time(); feat, lengths = ap.process_files(None, decoded=dec); torch.cuda.synchronize(); t_dev = time.time() - t0
    print("%-4s decode 32 x 10 s: %6.1f ms   upload + resample + front end: %6.1f ms   (frames %d)" % (name, t_dec * 1e3, t_dev * 1e3, lengths[0]))
