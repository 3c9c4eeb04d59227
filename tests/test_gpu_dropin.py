"""GPU tests of the drop-in class surface (models.AcousticModel / util.audioprocessor) driven the
way the reference's stt.py drives it, plus the data-parallel == gradient-accumulation equivalence."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import model as om  # noqa: E402  (checker only)


def synth(seed, n, sr=16000):
    rng = np.random.RandomState(seed)
    t = np.arange(n) / float(sr)
    return (0.1 * rng.randn(n) + 0.3 * np.sin(2 * np.pi * 300 * (1 + seed % 5) * t)).astype(np.float32)


def test_train_loop_surface_and_checkpoint(tmp_path):
    from models.AcousticModel import AcousticModel, Session
    from models.SpeechRecognizer import SpeechRecognizer
    cm = SpeechRecognizer("english").get_char_map()
    T, U, B = 60, 12, 2
    items = [[(synth(i, 16000 // 2 + 37 * i), 16000), txt, None]
             for i, txt in enumerate(["hello there", "it'll do", "good bye", "yes", "no way"])]
    model = AcousticModel(1, 32, B, T, U, 20, False, len(cm))
    sess = Session()
    train = model.build_dataset(items, B, T, U, "mfcc", cm)
    test = model.build_dataset(items[:2], B, T, U, "mfcc", cm)
    t_it, v_it = model.add_datasets_input(train, test)
    sess.run(t_it.initializer)
    sess.run(v_it.initializer)
    model.create_training_rnn(0.8, 0.5, 1, 1e-3, 0.33, use_iterator=True)
    model.initialize(sess)
    model.restore(sess, str(tmp_path / "acoustic"))
    before = model.engine.params.clone()
    loss, err, step, empty = model.run_train_step(sess, 2, 0.25)
    assert step == 1 and not empty and np.isfinite(loss) and 0 <= err
    assert not torch.equal(before, model.engine.params)
    loss, err, step, empty = model.run_train_step(sess, 2, 1.0)     # 5 items / batch 2 -> 3 mini-batches: runs dry
    assert step == 2 and empty
    sess.run(t_it.initializer)
    mloss, merr, estep = model.run_evaluation(sess)
    assert estep == 2 and np.isfinite(mloss)
    lr0 = model.learning_rate_var.eval()
    sess.run(model.learning_rate_decay_op)
    assert abs(model.learning_rate_var.eval() - 0.33 * lr0) < 1e-12
    model.save(sess, str(tmp_path / "acoustic"))
    clone = AcousticModel(1, 32, B, T, U, 20, False, len(cm))
    clone.create_forward_rnn()
    clone.restore(None, str(tmp_path / "acoustic"))
    assert torch.equal(clone.engine.params, model.engine.params) and clone.global_step.eval() == 2
    feats, n = clone_features(items[0][0][0])
    pred = clone.process_input(None, feats, [n, 0])
    assert pred.shape[0] == B and pred.dtype == np.int32


def clone_features(sig):
    from util.audioprocessor import AudioProcessor
    ap = AudioProcessor(60, "mfcc")
    feat, n = ap.process_signal(sig, 16000)
    x = np.zeros((60, 2, 20), np.float32)
    x[:len(feat), 0] = feat
    return x, min(n, 60)


@pytest.mark.parametrize("H", [32, 128], ids=["step-kernels", "dataflow-kernels"])
def test_data_parallel_equals_mini_batch_accumulation(H):
    """N ranks x batch b == one rank with mini_batch_size = N.  Two engine replicas play
    two ranks, their flat gradients are summed as the all-reduce would, and the result must equal
    both the single-engine accumulation and the oracle's train_step."""
    from rnn_speech_amd.engine import Engine
    L, D, C, B, T, U = 2, 8, 80, 3, 14, 5
    rng = np.random.RandomState(0)
    batches = []
    for r in range(2):
        x = rng.randn(T, B, D).astype(np.float32)
        ln = rng.randint(8, T + 1, size=B).astype(np.int32)
        dn = np.zeros((B, U), np.int32)
        dn[:, 0] = rng.randint(1, 79, size=B); dn[:, 1] = rng.randint(1, 79, size=B); dn[:, 2] = 79
        batches.append((x, ln, dn))
    ranks = [Engine(L, H, D, C, B, T, U, seed=9) for _ in range(2)]
    for eng, (x, ln, dn) in zip(ranks, batches):
        eng.zero_grads()
        eng.mini_batch(torch.as_tensor(x).cuda(), torch.as_tensor(ln).cuda(), torch.as_tensor(dn).cuda())
    reduced = ranks[0].grads + ranks[1].grads                 # all-reduce(SUM)
    single = Engine(L, H, D, C, B, T, U, seed=9)
    single.zero_grads()
    for x, ln, dn in batches:
        single.mini_batch(torch.as_tensor(x).cuda(), torch.as_tensor(ln).cuda(), torch.as_tensor(dn).cuda())
    scale = float(single.grads.abs().max().cpu())
    assert float((reduced - single.grads).abs().max().cpu()) < 1e-5 * scale
    for eng in ranks:
        eng.grads.copy_(reduced)
        eng.apply(3e-4, 1.0)
    single.apply(3e-4, 1.0)
    assert torch.equal(ranks[0].params, ranks[1].params)      # replicas stay bit-identical
    p = {k: v.astype(np.float64) for k, v in Engine(L, H, D, C, B, T, U, seed=9).to_numpy().items()}
    m = {k: np.zeros_like(v) for k, v in p.items()}
    v = {k: np.zeros_like(vv) for k, vv in p.items()}
    p0 = {k: vv.copy() for k, vv in p.items()}
    _, acc, _ = om.train_step(p, m, v, 1, [(x.astype(np.float64), ln, dn) for x, ln, dn in batches], L, 3e-4, 1.0,
                                carry_state=False)
    got = ranks[0].to_numpy()
    for k in p:
        # the first Adam step is lr * g/(|g| + eps_hat) = +-lr: only entries whose gradient sign is
        # numerically determined can be compared (a ~0 gradient flips sign between f32 and f64)
        sure = np.abs(acc[k]) > 1e-3 * np.abs(acc[k]).max()
        assert np.abs((got[k] - p0[k]) - (p[k] - p0[k]))[sure].max() < 0.05 * 3e-4, k


def test_dataset_pipeline_prefetch_cache_and_buckets(tmp_path):
    """File-backed datasets: the prefetching producer, the host feature cache (second epoch identical to the
    first without touching the files) and length-bucketed ordering."""
    import os
    import sys
    import wave
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from flac_writer import write_flac
    from rnn_speech_amd.acoustic_model import AcousticModel, bucketed_order
    from rnn_speech_amd.labels import ENGLISH_CHAR_MAP
    rng = np.random.RandomState(2)
    items = []
    for i in range(7):
        sr = 16000
        n = int(sr * (0.5 + 0.07 * ((i * 3) % 7)))
        t = np.arange(n) / float(sr)
        x = np.round((0.3 * np.sin(2 * np.pi * (200 + 60 * i) * t) + 0.05 * rng.randn(n)) * 20000).astype(np.int64)
        path = str(tmp_path / ("u%d.%s" % (i, "flac" if i % 2 else "wav")))
        if i % 2:
            write_flac(path, x, sr, 16, blocksize=4096, plan=[{"kind": "fixed2", "porder": 2}])
        else:
            with wave.open(path, "wb") as w:
                w.setnchannels(1)
                w.setsampwidth(2)
                w.setframerate(sr)
                w.writeframes(x.astype("<i2").tobytes())
        items.append([path, "utterance %d" % i, n / float(sr)])

    def run(ds):
        return [(f.cpu().numpy().copy(), l.copy(), d.copy()) for f, l, d in ds.batches()]

    args = (3, 150, 20, "mfcc", ENGLISH_CHAR_MAP)
    sync = run(AcousticModel.build_dataset(items, *args, n_mfcc=40, prefetch=0))
    assert len(sync) == 3 and sync[2][1][1] == 0 and sync[2][1][2] == 0          # 7 items, batch 3: padded tail
    ahead = run(AcousticModel.build_dataset(items, *args, n_mfcc=40, prefetch=2))
    cached_ds = AcousticModel.build_dataset(items, *args, n_mfcc=40, prefetch=2, feature_cache_mb=64)
    first = run(cached_ds)
    for p in [it[0] for it in items]:
        os.rename(p, p + ".gone")                                               # epoch 2 must not need the files
    second = run(cached_ds)
    reordered = cached_ds.with_items(bucketed_order(items, 3, rng=np.random.RandomState(0)))
    third = run(reordered)
    for other in (ahead, first, second):
        for (f0, l0, d0), (f1, l1, d1) in zip(sync, other):
            assert np.array_equal(l0, l1) and np.array_equal(d0, d1) and np.abs(f0 - f1).max() < 1e-5
    # bucketed: every batch holds neighbours in duration; same multiset of utterances as before
    order = bucketed_order(items, 3, rng=np.random.RandomState(0))
    durs = [[it[2] for it in order[i:i + 3]] for i in range(0, 7, 3)]
    spans = sorted((min(d), max(d)) for d in durs)
    assert all(spans[i][1] <= spans[i + 1][0] for i in range(len(spans) - 1))
    assert sorted(int(l) for b in third for l in b[1] if l > 0) == sorted(int(l) for b in sync for l in b[1] if l > 0)
    with pytest.raises(Exception):
        run(AcousticModel.build_dataset(items, *args, n_mfcc=40, prefetch=2))   # files are gone: decode error surfaces
