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


def test_process_input_values_and_default_beam_decoder():
    """process_input (the reference's forward-only path, :705-721): logits against the float64 oracle, predictions =
    width-100 prefix beam search + merge_repeated (the reference's decoder, :312-314) -- not just shapes."""
    from models.AcousticModel import AcousticModel
    from rnn_speech_amd import ops
    L, H, D, C, B, T, U = 2, 128, 20, 80, 3, 40, 10
    model = AcousticModel(L, H, B, T, U, D, False, C)
    model.create_forward_rnn()
    assert model.decoder == "beam" and model.beam_width == 100 and model.merge_repeated
    rng = np.random.RandomState(2)
    p = model.engine.to_numpy()
    p["output_w"] = (rng.randn(H, C) * 0.8).astype(np.float32)         # peaky logits: non-trivial decodes
    model.engine.load_numpy(p)
    x = rng.randn(T, B, D).astype(np.float32)
    lens = np.array([40, 23, 0], np.int32)
    pred = model.process_input(None, x, lens)
    p64 = {k: v.astype(np.float64) for k, v in p.items()}
    logits_ref, _, _ = om.forward(p64, x.astype(np.float64), lens, L)
    got = model.engine.logits.cpu().numpy()
    assert np.abs(got - logits_ref).max() < 1e-4 * np.abs(logits_ref).max()
    ids, out_len, _ = ops.ctc_beam_search(logits_ref.astype(np.float32), lens, 100, True)
    assert out_len[0] > 0 and out_len[2] == 0
    for b in range(B):
        row = pred[b]
        assert list(row[:out_len[b]]) == list(ids[b, :out_len[b]])
        assert np.all(row[out_len[b]:] == C)                           # padded with num_labels (:718)
    model.decoder = "greedy"
    pred_g = model.process_input(None, x, lens)
    ref_g = om.greedy_decode(logits_ref, lens)
    for b in range(B):                                                  # greedy + merge_repeated
        want = [k for i, k in enumerate(ref_g[b]) if i == 0 or k != ref_g[b][i - 1]]
        assert [int(v) for v in pred_g[b] if v != C] == want


@pytest.mark.parametrize("lag", [0, 2])
def test_training_time_beam_decoder_runs_asynchronously_with_the_reference_values(lag):
    """train_decoder = "beam": the reference's decoder (width 100 + merge_repeated, models/AcousticModel.py:312-314) on every
    TRAINING mini-batch, off the training thread.  The error rate of every mini-batch is checked by value (host beam search on
    that step's logits + the oracle's edit distance, mean of distance / len(truth), :370) and by position: with lag = 0 step k
    reports mini-batch k, with lag = 2 it reports mini-batch k-2 (the first step waits for its own decode, the next two repeat it), and draining
    returns the rest.  Losses are those of the greedy-decoding model (the decoder only feeds the logged scalar)."""
    from models.AcousticModel import AcousticModel, Session
    from rnn_speech_amd import ops
    L, H, D, C, B, T, U = 2, 128, 20, 80, 4, 50, 12
    rng = np.random.RandomState(3)
    batches = []
    for i in range(6):
        x = rng.randn(T, B, D).astype(np.float32)
        lens = rng.randint(20, T + 1, size=B).astype(np.int32)
        dense = np.zeros((B, U), np.int32)
        for b in range(B):
            n = rng.randint(2, 8)
            dense[b, :n] = rng.randint(1, C - 1, size=n)
            dense[b, n] = C - 1
        batches.append((x, lens, dense))

    def make(decoder):
        m = AcousticModel(L, H, B, T, U, D, False, C)
        m.create_training_rnn(1.0, 1.0, 1.0, 1e-3, 0.5)
        p = m.engine.to_numpy()
        p["output_w"] = (np.random.RandomState(9).randn(H, C) * 0.8).astype(np.float32)      # peaky: non-trivial decodes
        m.engine.load_numpy(p)
        m.train_decoder, m.train_decoder_lag = decoder, lag
        return m
    model, ref = make("beam"), make("greedy")
    got, want, losses, ref_losses = [], [], [], []
    for x, lens, dense in batches:
        model.feed(x, lens, dense)
        loss, err, _, _ = model.run_train_step(Session(), 1, 1.0)
        got.append(err); losses.append(loss)
        # what the reference computes for THIS mini-batch, from the logits the step left behind
        ids, out_len, _ = ops.ctc_beam_search(model.engine.logits, torch.as_tensor(lens), 100, True)
        rates = []
        for b in range(B):
            truth = [int(v) for v in dense[b] if v != 0]
            rates.append(om.edit_distance(list(ids[b, :out_len[b]]), truth) / float(len(truth)))
        want.append(float(np.mean(rates)))
        ref.feed(x, lens, dense)
        ref_losses.append(ref.run_train_step(Session(), 1, 1.0)[0])
    np.testing.assert_allclose(losses, ref_losses, rtol=1e-6)
    assert max(want) > 0 and len(set(np.round(want, 6))) > 2                    # (distinct values: positions can be told apart)
    if lag == 0:
        np.testing.assert_allclose(got, want, rtol=1e-12)
    else:
        # the first step waits for its own decode, the next `lag` steps have nothing due and repeat it; from then on a pure delay
        np.testing.assert_allclose(got[:lag + 1], [want[0]] * (lag + 1), rtol=1e-12)
        np.testing.assert_allclose(got[lag + 1:], want[1:len(want) - lag], rtol=1e-12)
    rest = model._async_beam.collect(drain=True)
    np.testing.assert_allclose(rest, want[len(want) - lag:] if lag else [], rtol=1e-12)
    model._async_beam.close()


def test_evaluate_full_wer_cer_by_value():
    """evaluate_full (reference models/AcousticModel.py:723-777): WER and CER as VALUES -- features from the oracle front end,
    logits from the float64 oracle, width-100 beam + merge_repeated on those logits, the reference-pinned label codec and
    calculate_wer / calculate_cer (tests/golden/wer_cer.json), averaged the reference's way (per-utterance rates, x100); incl. a
    sample that is skipped as too long and the zero-padded final batch."""
    from models.AcousticModel import AcousticModel
    from oracle import frontend as ofe
    from oracle import labels as olab
    from rnn_speech_amd import ops
    L, H, D, C, B, T, U = 2, 128, 20, 80, 2, 60, 30
    model = AcousticModel(L, H, B, T, U, D, False, C)
    model.create_forward_rnn()
    rng = np.random.RandomState(5)
    p = model.engine.to_numpy()
    p["output_w"] = (rng.randn(H, C) * 0.8).astype(np.float32)         # peaky posteriors: long, non-trivial decodes
    model.engine.load_numpy(p)
    sr = 16000

    def sig(seed, n):
        r = np.random.RandomState(seed)
        t = np.arange(n) / float(sr)
        return (0.1 * r.randn(n) + 0.3 * np.sin(2 * np.pi * (300 + 40 * seed) * t)).astype(np.float32)
    texts = ["hello there", "good bye it'll do", "yes", "no way", "well being now"]
    items = [((sig(i, 4000 + 700 * i), sr), texts[i], 0.5) for i in range(5)]
    items.insert(2, ((sig(9, 160 * 80), sr), "too long input", 0.5))          # 81 frames > T: skipped with a warning
    char_map = olab.CHAR_MAP
    wer, cer = model.evaluate_full(None, items, T, "mfcc", char_map, n_mfcc=D)

    p64 = {k: v.astype(np.float64) for k, v in p.items()}
    wl, cl = [], []
    kept = [it for it in items if it[1] != "too long input"]
    for i in range(0, len(kept), B):
        chunk = kept[i:i + B]
        x = np.zeros((T, B, D))
        lens = np.zeros(B, np.int32)
        for b, ((s_, r_), _, _) in enumerate(chunk):
            f = ofe.mfcc(s_, r_, n_mfcc=D)
            x[:len(f), b] = f
            lens[b] = len(f)
        logits, _, _ = om.forward(p64, x, lens, L)
        ids, out_len, _ = ops.ctc_beam_search(logits.astype(np.float32), lens, 100, True)
        for b, (_, truth, _) in enumerate(chunk):
            hyp = olab.labels_to_str(char_map, [int(v) for v in ids[b, :out_len[b]]])
            wl.append(om.calculate_wer(hyp, truth) / float(len(truth.split())))
            cl.append(om.calculate_cer(hyp, truth) / float(len(truth.replace(" ", ""))))
    assert len(wl) == 5 and max(cl) > 0
    assert abs(wer - 100.0 * sum(wl) / len(wl)) < 1e-9 and abs(cer - 100.0 * sum(cl) / len(cl)) < 1e-9, (wer, cer, wl, cl)


def test_c_abi_communicator_single_rank():
    """amdspeech_comm_* / amdspeech_allreduce_sum_f32 / amdspeech_broadcast_f32 (RCCL behind the C ABI): a world of
    one rank on this box -- the id, the communicator and both collectives must work and leave the buffer as is
    (the multi-rank arithmetic is covered by the gloo tests; the 8-GPU run is the driver's)."""
    import ctypes as C_
    from rnn_speech_amd import lib as _l
    lib = _l.load()
    ident = (C_.c_char * _l.COMM_ID_BYTES)()
    _l.check(lib.amdspeech_comm_unique_id(ident), "comm_unique_id")
    comm = C_.c_void_p()
    _l.check(lib.amdspeech_comm_init(ident, 0, 1, C_.byref(comm)), "comm_init")
    buf = torch.randn(6359632 + 64, device="cuda")                     # the cfg2 flat gradient size
    ref = buf.clone()
    stream = C_.c_void_p(torch.cuda.current_stream().cuda_stream)
    _l.check(lib.amdspeech_allreduce_sum_f32(comm, stream, C_.c_void_p(buf.data_ptr()), buf.numel()), "allreduce")
    _l.check(lib.amdspeech_broadcast_f32(comm, stream, C_.c_void_p(buf.data_ptr()), buf.numel(), 0), "broadcast")
    torch.cuda.synchronize()
    assert torch.equal(buf, ref)
    assert lib.amdspeech_broadcast_f32(comm, stream, C_.c_void_p(buf.data_ptr()), buf.numel(), 3) != 0   # bad root
    _l.check(lib.amdspeech_comm_destroy(comm), "comm_destroy")


_SHARE_GPU_WORKER = r"""
import os, sys, numpy as np, torch
sys.path.insert(0, %(root)r)
os.environ["AMDSPEECH_FLOW"] = "0"            # two ranks time-slice ONE GPU here: the dataflow kernels need it alone
os.environ["AMDSPEECH_SHARE_GPU"] = "1"
os.environ["AMDSPEECH_DIST_BACKEND"] = "gloo"
torch.cuda.set_device(0)
from rnn_speech_amd import dataparallel
from models.AcousticModel import AcousticModel, Session
from models.SpeechRecognizer import SpeechRecognizer
grp = dataparallel.current()
rank, world = grp.rank, grp.world
cm = SpeechRecognizer("english").get_char_map()
T, U, B = 60, 12, 2
def synth(seed, n, sr=16000):
    rng = np.random.RandomState(seed)
    t = np.arange(n) / float(sr)
    return (0.1 * rng.randn(n) + 0.3 * np.sin(2 * np.pi * 300 * (1 + seed %% 5) * t)).astype(np.float32)
texts = ["hello there", "it'll do", "good bye", "yes", "no way", "well", "so long", "fine", "okay then", "right"]
items = [[(synth(i, 16000 // 2 + 37 * i), 16000), texts[i], None] for i in range(10)]
mine = items[:6] if rank == 0 else items[6:]          # UNEQUAL shards on purpose: 3 vs 2 mini-batches
model = AcousticModel(1, 32, B, T, U, 20, False, len(cm))
sess = Session()
t_it, v_it = model.add_datasets_input(model.build_dataset(mine, B, T, U, "mfcc", cm),
                                      model.build_dataset(items[:2], B, T, U, "mfcc", cm))
sess.run(t_it.initializer); sess.run(v_it.initializer)
model.create_training_rnn(1.0, 1.0, 1, 1e-3, 0.33, use_iterator=True)
model.engine.params.add_(0.01 * rank)                 # replicas differ until restore() broadcasts rank 0's
model.restore(sess, %(ckpt)r)
steps = 0
for epoch in range(2):
    while True:
        loss, err, step, empty = model.run_train_step(sess, 1, 1.0)
        if empty:
            break
        steps += 1
    sess.run(t_it.initializer)
assert steps == 4 and model.global_step.eval() == 4, (steps, model.global_step.eval())      # 2 per epoch: the shorter shard
model.save(sess, %(ckpt)r)
flat = model.engine.params.double().cpu()
chk = [float(flat.sum()), float((flat * flat).sum()), float(loss), float(err)]
mean = [v / world for v in grp.sum_scalars(chk)]
assert all(a == b or abs(a - b) <= 1e-9 * abs(b) for a, b in zip(chk, mean)), (chk, mean)
assert np.isfinite(loss)
if rank == 0:
    z = np.load(os.path.join(%(ckpt)r, "acousticmodel.ckpt-4.npz"))
    assert int(z["adam/step"]) == 4 and "adam/m/Input_Layer/input_w" in z.files
print("rank", rank, "ok")
"""


def test_data_parallel_drop_in_loop_on_the_real_engine(tmp_path):
    """Two ranks (gloo, both on this box's one GPU, launch-per-diagonal kernels) drive AcousticModel.run_train_step
    with the REAL engine through two epochs with unequal shards: same step count everywhere, no hang, bit-identical
    replicas and identical logged scalars, rank 0 alone writes the checkpoint (with the Adam moments)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "share_gpu_worker.py"
    script.write_text(_SHARE_GPU_WORKER % {"root": root, "ckpt": str(tmp_path / "acoustic")})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29553", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29553", str(script)],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert out.stdout.count("ok") == 2


_BN_DP_WORKER = r"""
import os, sys, numpy as np, torch
sys.path.insert(0, %(root)r)
sys.path.insert(0, os.path.join(%(root)r, "tests"))
os.environ["AMDSPEECH_FLOW"] = "0"
os.environ["AMDSPEECH_SHARE_GPU"] = "1"
os.environ["AMDSPEECH_DIST_BACKEND"] = "gloo"
torch.cuda.set_device(0)
from rnn_speech_amd import dataparallel
from rnn_speech_amd.engine import Engine
from oracle import model as om
from test_gpu_model import make_batch, rel_err
grp = dataparallel.current()
rank, world = grp.rank, grp.world
L, H, D, C, Bl, T, U = 2, 32, 12, 80, 3, 18, 7
B = Bl * world
x, lengths, dense = make_batch(T, B, D, C, U, seed=77)
if rank == 1:
    lengths[Bl:] = np.minimum(lengths[Bl:], T - 5)      # rank 1's longest utterance is shorter: it still supplies all T frames
lengths = np.asarray(grp.broadcast_object(lengths, 1))
sl = slice(rank * Bl, (rank + 1) * Bl)
# default: every rank normalises ITS mini-batch with its own moments -- the reference's mini_batch_size = world accumulation
# (each mini-batch graph takes tf.nn.moments of its own batch, :253-259): the all-reduced gradient is the SUM of the two
# 3-utterance oracles' gradients, and no collective runs inside the step (max_len keeps its early stop)
loc = Engine(L, H, D, C, Bl, T, U, seed=21, normalization=True)
assert not loc.sync_batch_norm
p64 = {k: v.astype(np.float64) for k, v in loc.to_numpy().items()}
g_sum = None
for r in range(world):
    s_r = slice(r * Bl, (r + 1) * Bl)
    lg_r, _, cache_r = om.forward(p64, x[:, s_r].astype(np.float64), lengths[s_r], L, keep_cache=True, normalization=True)
    loss_r, dl_r = om.ctc_loss_and_grad(lg_r, om.sparsify_labels(dense[s_r], C), lengths[s_r])
    g_r = om.backward(p64, cache_r, dl_r, lengths[s_r], L)
    g_sum = g_r if g_sum is None else {k: g_sum[k] + g_r[k] for k in g_r}
    if r == rank:
        lg_mine, loss_mine = lg_r, loss_r
loc.zero_grads()
loc.mini_batch(torch.as_tensor(x[:, sl]).cuda(), torch.as_tensor(lengths[sl]).cuda(), torch.as_tensor(dense[sl]).cuda(),
               max_len=int(lengths[sl].max()))
assert loc._Tr == int(lengths[sl].max())
loc.all_reduce_grads()
torch.cuda.synchronize()
Tv = int(lengths[sl].max())
assert rel_err(loc.logits.cpu().numpy()[:Tv], lg_mine[:Tv]) < 1e-4
np.testing.assert_allclose(loc.loss.cpu().numpy(), loss_mine, rtol=1e-3, atol=1e-5)
g = loc.to_numpy(loc.grads)
for k in g_sum:
    if k == "input_b":
        assert np.abs(g[k]).max() < 1e-4 * np.abs(g["input_w"]).max()
        continue
    assert rel_err(g[k], g_sum[k]) < 2e-3, k
# opt-in sync_batch_norm (a deviation from the reference): the moments span the ranks' batches
eng = Engine(L, H, D, C, Bl, T, U, seed=21, normalization=True, sync_batch_norm=True)
logits_ref, _, cache = om.forward(p64, x.astype(np.float64), lengths, L, keep_cache=True, normalization=True)
loss_ref, dl_ref = om.ctc_loss_and_grad(logits_ref, om.sparsify_labels(dense, C), lengths)
g_ref = om.backward(p64, cache, dl_ref, lengths, L)
eng.zero_grads()
eng.mini_batch(torch.as_tensor(x[:, sl]).cuda(), torch.as_tensor(lengths[sl]).cuda(), torch.as_tensor(dense[sl]).cuda(),
               max_len=int(lengths[sl].max()))
eng.all_reduce_grads()
torch.cuda.synchronize()
Tv = int(lengths[sl].max())
assert rel_err(eng.logits.cpu().numpy()[:Tv], logits_ref[:Tv, sl]) < 1e-4
np.testing.assert_allclose(eng.loss.cpu().numpy(), loss_ref[sl], rtol=1e-3, atol=1e-5)
g = eng.to_numpy(eng.grads)
for k in g_ref:
    if k == "input_b":
        assert np.abs(g[k]).max() < 1e-4 * np.abs(g["input_w"]).max()
        continue
    assert rel_err(g[k], g_ref[k]) < 2e-3, k
print("rank", rank, "ok")
"""


def test_batch_normalization_under_data_parallelism(tmp_path):
    """batch_normalization under data parallelism, two ranks (sharing this box's GPU), 3 utterances each.  Default: per-rank
    moments = the reference's gradient accumulation over two mini-batches (sum of the two 3-utterance oracles).  Opt-in
    sync_batch_norm: the oracle's logits, losses and summed gradients for the 6-utterance batch -- the moments and both
    backward sums cross the ranks (amdspeech_batchnorm_sum / _apply / _bwd_sums / _bwd_apply + all-reduce)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "bn_dp_worker.py"
    script.write_text(_BN_DP_WORKER % {"root": root})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29557", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29557", str(script)],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert out.stdout.count("ok") == 2


def test_second_stream_work_beside_a_dataflow_step_never_hangs():
    """INTEGRATION.md says nothing may be launched beside the dataflow LSTM kernels (one resident workgroup per CU,
    spinning on its siblings).  If a caller does it anyway -- a 256-CU-filling GEMM loop on another stream during the
    step -- the contract is: either correct results (the late workgroups were scheduled when the filler drained) or a
    clean AmdSpeechError from Engine.check(); never a hang, never silently wrong numbers."""
    from rnn_speech_amd import ops
    from rnn_speech_amd.engine import Engine
    from rnn_speech_amd.lib import AmdSpeechError
    L, H, D, C, B, T, U = 3, 512, 40, 80, 32, 200, 40
    eng = Engine(L, H, D, C, B, T, U, seed=5)
    rng = np.random.RandomState(0)
    x = torch.as_tensor(rng.randn(T, B, D).astype(np.float32)).cuda()
    lens = torch.full((B,), T, dtype=torch.int32).cuda()
    dense = np.zeros((B, U), np.int32)
    dense[:, :10] = rng.randint(1, C - 1, size=(B, 10)); dense[:, 10] = C - 1
    dlab = torch.as_tensor(dense).cuda()
    with eng.on_stream():
        eng.zero_grads()
        eng.mini_batch(x, lens, dlab)
    torch.cuda.synchronize()
    eng.check()
    ref_loss, ref_grads = eng.loss.clone(), eng.grads.clone()
    a = torch.randn(4096, 4096, device="cuda"); bmat = torch.randn(4096, 4096, device="cuda")
    out = torch.empty(4096, 4096, device="cuda")
    side = torch.cuda.Stream()
    clean = raised = 0
    for trial in range(3):
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            for _ in range(6 + 6 * trial):                     # ~1 ms each: overlaps forward and backward kernels
                ops.gemm(a, bmat, out=out)
        with eng.on_stream():
            eng.zero_grads()
            eng.mini_batch(x, lens, dlab)
        torch.cuda.synchronize()                               # returns: no hang
        try:
            eng.check()
        except AmdSpeechError:
            raised += 1
            continue
        clean += 1
        assert float((eng.loss - ref_loss).abs().max()) <= 1e-4 * float(ref_loss.abs().max())
        assert float((eng.grads - ref_grads).abs().max()) <= 2e-4 * float(ref_grads.abs().max())
    assert clean + raised == 3


def test_timeline_writes_chrome_traces(tmp_path):
    """stt.py --timeline (reference `_write_timeline`, :873-885): `timeline-step-<i>.ctf.json` and `timeline-end_batch.ctf.json`
    in the tensorboard directory, chrome trace format, with the GPU stages of the step as complete events."""
    import json
    from models.AcousticModel import AcousticModel, Session
    from models.SpeechRecognizer import SpeechRecognizer
    cm = SpeechRecognizer("english").get_char_map()
    T, U, B = 60, 12, 2
    rng = np.random.RandomState(0)
    items = [[(0.1 * rng.randn(9000).astype(np.float32), 16000), t, None] for t in ("hello there", "good bye", "yes", "no way")]
    model = AcousticModel(1, 32, B, T, U, 20, False, len(cm))
    sess = Session()
    t_it, v_it = model.add_datasets_input(model.build_dataset(items, B, T, U, "mfcc", cm), model.build_dataset(items[:2], B, T, U, "mfcc", cm))
    sess.run(t_it.initializer)
    model.create_training_rnn(1.0, 1.0, 1, 1e-3, 0.33, use_iterator=True)
    model.add_tensorboard(sess, str(tmp_path), None, timeline_enabled=True)
    model.run_train_step(sess, 2, 1.0)
    for name in ("timeline-step-0.ctf.json", "timeline-step-1.ctf.json", "timeline-end_batch.ctf.json"):
        ev = json.load(open(tmp_path / name))["traceEvents"]
        spans = [e for e in ev if e["ph"] == "X" and e["pid"] == 0]
        assert spans and all(e["dur"] >= 0 for e in spans), name
    names = [e["name"] for e in json.load(open(tmp_path / "timeline-step-0.ctf.json"))["traceEvents"] if e.get("pid") == 0 and e["ph"] == "X"]
    assert names == ["forward", "ctc", "backward"]


def test_input_pipeline_staging_pool_survives_iterator_resets():
    """The dataset's producer thread packs waveforms into pinned blocks ahead of their upload; an iterator that is reset in
    mid-epoch hands its unused blocks back (the pool neither leaks nor grows), and the batches of a fresh epoch are the same."""
    from models.AcousticModel import AcousticModel, Session
    from models.SpeechRecognizer import SpeechRecognizer
    from rnn_speech_amd import audioprocessor as ap
    cm = SpeechRecognizer("english").get_char_map()
    T, U, B = 60, 12, 2
    rng = np.random.RandomState(1)
    items = [[(0.1 * rng.randn(8000 + 100 * i).astype(np.float32), 16000), "yes no", None] for i in range(12)]
    model = AcousticModel(1, 32, B, T, U, 20, False, len(cm))
    sess = Session()
    t_it, v_it = model.add_datasets_input(model.build_dataset(items, B, T, U, "mfcc", cm), model.build_dataset(items[:2], B, T, U, "mfcc", cm))
    first = None
    for epoch in range(8):
        sess.run(t_it.initializer)
        feat, lengths, dense = t_it.get_next()
        torch.cuda.synchronize()
        if first is None:
            first = (feat.clone(), lengths.copy(), dense.copy())
        else:
            assert torch.equal(feat, first[0]) and (lengths == first[1]).all() and (dense == first[2]).all()
        t_it.get_next()                                  # ... and abandon the epoch after two of six mini-batches
    torch.cuda.synchronize()
    assert len(ap._PINNED._blocks) <= ap._PINNED._limit
    claimed = sum(1 for b in ap._PINNED._blocks if b.claimed)
    assert claimed <= 4, claimed                         # at most what the live producer has prepared ahead
