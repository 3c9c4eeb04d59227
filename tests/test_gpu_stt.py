"""End-to-end run of the reference's command-line surface (stt.py --train_acoustic / --evaluate / --file)
on a tiny synthetic corpus of WAV files: config.ini keys, manifests, dataset batching, training loop with
checkpoints and the plateau rule, evaluation (WER/CER) and single-file transcription."""
import os
import sys
import wave

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CONFIG = """
[acoustic_network_params]
num_layers : 1
hidden_size : 32
dropout_input_keep_prob : 0.8
dropout_output_keep_prob : 0.5
batch_size : 2
mini_batch_size : 2
learning_rate : 0.001
lr_decay_factor : 0.33
grad_clip : 1
signal_processing : mfcc
language : english
rnn_state_reset_ratio : 0.25

[general]
use_config_file_if_checkpoint_exists : True
steps_per_checkpoint : 2
steps_per_evaluation : 2
checkpoint_dir : %(dir)s/ckpt

[training]
training_dataset_dirs : %(dir)s/train.tsv
test_dataset_dirs : %(dir)s/test.tsv
max_input_seq_length : 60
max_target_seq_length : 20
batch_normalization : False
dataset_size_ordering : True

[logging]
log_level : WARNING
"""


def write_wav(path, seed, seconds=0.5, sr=22050):
    rng = np.random.RandomState(seed)
    t = np.arange(int(seconds * sr)) / float(sr)
    sig = 0.05 * rng.randn(len(t)) + 0.3 * np.sin(2 * np.pi * (200 + 50 * seed) * t)
    with wave.open(path, "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(sr)
        w.writeframes((np.clip(sig, -1, 1) * 32767).astype("<i2").tobytes())


def test_stt_train_evaluate_file(tmp_path, monkeypatch, capsys):
    d = str(tmp_path)
    texts = ["hello there", "good bye", "it'll do", "yes", "no way", "well-being"]
    with open(os.path.join(d, "train.tsv"), "w") as tr, open(os.path.join(d, "test.tsv"), "w") as te:
        for i, txt in enumerate(texts):
            p = os.path.join(d, "u%d.wav" % i)
            write_wav(p, i)
            (tr if i < 5 else te).write("%s\t%s\n" % (p, txt))
        te.write("%s\t%s\n" % (os.path.join(d, "u0.wav"), texts[0]))
    cfg = os.path.join(d, "config.ini")
    with open(cfg, "w") as fh:
        fh.write(CONFIG % {"dir": d})
    import stt
    monkeypatch.setattr(sys, "argv", ["stt.py", "--train_acoustic", "--config", cfg, "--max_epoch", "1"])
    stt.main()
    ckpt = os.path.join(d, "ckpt", "acoustic")
    assert os.path.exists(os.path.join(ckpt, "checkpoint"))
    assert any(f.endswith(".npz") for f in os.listdir(ckpt))
    assert os.path.exists(os.path.join(d, "ckpt", "hyperparams.p"))
    monkeypatch.setattr(sys, "argv", ["stt.py", "--evaluate", "--config", cfg])
    stt.main()
    out = capsys.readouterr().out
    assert "Resulting WER" in out and "Resulting CER" in out
    monkeypatch.setattr(sys, "argv", ["stt.py", "--file", os.path.join(d, "u1.wav"), "--config", cfg])
    stt.main()
    out = capsys.readouterr().out
    assert out.strip().startswith("[")
