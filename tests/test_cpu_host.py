"""CPU suite, part 2: host-side logic of the product (no GPU, no oracle in the product path):
label codec, config reader, parameter layout, dataset batching contract, the C-ABI library
(loads and exports every declared symbol), and the data-parallel gradient path on gloo."""
import ctypes
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_loads_and_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    from rnn_speech_amd import lib
    handle = lib.load()
    header = open(os.path.join(ROOT, "include", "amdspeech.h")).read()
    declared = set(re.findall(r"\b(amdspeech_[a-z0-9_]+)\s*\(", header))
    declared -= {"amdspeech_lstm_desc"}
    assert declared, "no declarations parsed"
    assert declared == set(lib.PROTOTYPES), (declared ^ set(lib.PROTOTYPES))
    for name in declared:
        assert getattr(handle, name) is not None
    assert handle.amdspeech_version() >= 100
    # pure host-side entry points may be called without a GPU
    d = lib.LstmDesc(1001, 32, 512, 3, 1.0, 1.0, 0)
    assert handle.amdspeech_lstm_workspace_bytes(ctypes.byref(d)) > 2 * 10 ** 9
    bad = lib.LstmDesc(10, 2, 50, 1, 1.0, 1.0, 0)          # H = 50 is not a multiple of 16
    assert handle.amdspeech_lstm_workspace_bytes(ctypes.byref(bad)) == 0
    assert b"multiple of 16" in handle.amdspeech_last_error()
    assert handle.amdspeech_frontend_num_frames(0, 160000, 16000) == 1001
    assert handle.amdspeech_frontend_num_frames(1, 160000, 16000) == 998
    assert handle.amdspeech_frontend_num_frames(0, 220500, 22050) == 1003
    assert handle.amdspeech_frontend_num_frames(1, 220500, 22050) == 1000
    assert handle.amdspeech_ctc_workspace_bytes(1001, 32, 80, 161) > 0


def test_product_path_fails_loudly_without_the_library(tmp_path, monkeypatch):
    from rnn_speech_amd import lib
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", str(tmp_path / "missing.so"))
    with pytest.raises(lib.AmdSpeechError):
        lib.load()


def test_product_does_not_import_the_oracle():
    """The oracle is test infrastructure: nothing under rnn-speech_amd/, models/, util/, stt.py may touch it."""
    offenders = []
    for base in ("rnn-speech_amd", "models", "util"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith(".py"):
                    src = open(os.path.join(dirpath, f)).read()
                    if re.search(r"^\s*(from|import)\s+oracle\b", src, re.M):
                        offenders.append(os.path.join(dirpath, f))
    src = open(os.path.join(ROOT, "stt.py")).read()
    if re.search(r"^\s*(from|import)\s+oracle\b", src, re.M):
        offenders.append("stt.py")
    assert not offenders, offenders


def test_product_label_codec_golden(golden_dir):
    from util.dataprocessor import DataProcessor
    from models.SpeechRecognizer import SpeechRecognizer
    cm = SpeechRecognizer("english").get_char_map()
    g = json.load(open(os.path.join(golden_dir, "labels.json")))
    assert g["char_map"] == cm and len(cm) == 80
    for e in g["encode"]:
        cleaned = DataProcessor.clean_label(e["text"])
        assert cleaned == e["cleaned"]
        assert DataProcessor.get_str_labels(cm, cleaned) == e["ids"]
        assert DataProcessor.get_labels_str(cm, e["ids"]) == e["decoded"]
    for d in g["decode"]:
        assert DataProcessor.get_labels_str(cm, d["ids"]) == d["decoded"]
    # the reference's own unit-test answers (util/test_dataProcessor.py:139-149)
    assert DataProcessor.get_str_labels(cm, DataProcessor.clean_label("it'll")) == [60, 45, 1, 79]
    assert DataProcessor.get_str_labels(cm, DataProcessor.clean_label("'d")) == [0, 79]
    onehot = DataProcessor.get_str_to_one_hot_encoded(cm, "ab", add_eos=False)
    assert [int(v.argmax()) for v in onehot] == [52, 27] and onehot[0].sum() == 1


def test_wer_cer_product(golden_dir):
    from rnn_speech_amd.acoustic_model import AcousticModel, _edit_distance
    for w in json.load(open(os.path.join(golden_dir, "wer_cer.json"))):
        assert AcousticModel.calculate_wer(w["a"], w["b"]) == w["wer"]
        assert AcousticModel.calculate_cer(w["a"], w["b"]) == w["cer"]
    rng = np.random.RandomState(0)
    for _ in range(100):
        a = rng.randint(0, 4, size=rng.randint(0, 10))
        b = rng.randint(0, 4, size=rng.randint(0, 10))
        # brute-force reference: classic O(nm) table
        d = np.zeros((len(a) + 1, len(b) + 1), int)
        d[:, 0] = np.arange(len(a) + 1)
        d[0, :] = np.arange(len(b) + 1)
        for i in range(1, len(a) + 1):
            for j in range(1, len(b) + 1):
                d[i, j] = min(d[i - 1, j - 1] + (a[i - 1] != b[j - 1]), d[i - 1, j] + 1, d[i, j - 1] + 1)
        assert _edit_distance(a, b) == d[-1, -1]


def test_config_reader_reads_the_reference_config(golden_dir, tmp_path):
    """Same keys, same parsing as the reference handler on the reference's own config.ini content."""
    from util.hyperparams import read_config_file, HyperParameterHandler
    ini = json.load(open(os.path.join(golden_dir, "config_ini.json")))
    cfg = tmp_path / "config.ini"
    with open(cfg, "w") as fh:
        for section, kv in ini.items():
            fh.write("[%s]\n" % section)
            for k, v in kv.items():
                if k == "checkpoint_dir":
                    v = str(tmp_path / "ckpt")
                if k == "log_file":
                    continue
                fh.write("%s : %s\n" % (k, v))
    d = read_config_file(str(cfg))
    assert (d["num_layers"], d["hidden_size"], d["batch_size"], d["mini_batch_size"]) == (5, 1024, 10, 3)
    assert d["learning_rate"] == 0.0003 and d["lr_decay_factor"] == 0.33 and d["grad_clip"] == 1
    assert d["signal_processing"] == "fbank" and d["rnn_state_reset_ratio"] == 0.25
    assert d["max_input_seq_length"] == 3510 and d["max_target_seq_length"] == 600
    assert d["n_mfcc"] == 20 and d["batch_normalization"] is False
    h = HyperParameterHandler(str(cfg))
    assert os.path.exists(os.path.join(str(tmp_path / "ckpt"), "hyperparams.p"))
    assert not h.check_changed(h.get_hyper_params())
    changed = dict(h.get_hyper_params(), hidden_size=512)
    assert h.check_changed(changed)


def test_param_layout_matches_reference_checkpoint_shapes():
    """3x1024 / 120-dim / 80 labels has 25,384,016 parameters."""
    from rnn_speech_amd.engine import ParamLayout
    lay = ParamLayout(3, 1024, 120, 80)
    assert lay.num_params() == 25384016 - 0 or lay.num_params() == 25384014   # ckpt adds global_step + learning_rate
    assert lay.slots["kernel_0"][1] == (2048, 4096) and lay.slots["output_w"][1] == (1024, 80)
    assert all(off % 64 == 0 for off, _ in lay.slots.values())
    assert lay.kernel_stride == lay.slots["kernel_2"][0] - lay.slots["kernel_1"][0]
    assert ParamLayout(3, 512, 40, 80).num_params() == 6359632


_DP_WORKER = r"""
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %(root)r)
from oracle import model as om
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo")
L, H, D, C, B, T, U = 1, 8, 5, 80, 2, 10, 4
p = om.init_params(L, H, D, C, seed=1, dtype=np.float64)
rng = np.random.RandomState(7)
xs = [rng.randn(T, B, D) for _ in range(world)]
lens = [np.array([10, 7]) for _ in range(world)]
labs = []
for r in range(world):
    d = np.zeros((B, U), int); d[:, 0] = [3 + r, 9]; d[:, 1] = [5, 79]; d[0, 2] = 79
    labs.append(d)
def grads(x, ln, dn):
    lg, _, cache = om.forward(p, x, ln, L, keep_cache=True)
    _, dl = om.ctc_loss_and_grad(lg, om.sparsify_labels(dn, C), ln)
    return om.backward(p, cache, dl, ln, L)
from rnn_speech_amd.engine import ParamLayout
lay = ParamLayout(L, H, D, C)
flat = torch.zeros(lay.total, dtype=torch.float64)
mine = grads(xs[rank], lens[rank], labs[rank])
for k, v in mine.items():
    lay.view(flat, k).copy_(torch.as_tensor(v))
dist.all_reduce(flat, op=dist.ReduceOp.SUM)      # what Engine.all_reduce_grads does (one flat SUM all-reduce)
# oracle of the exchange: the reference's gradient accumulation over mini_batch_size = world mini-batches
acc = None
for r in range(world):
    g = grads(xs[r], lens[r], labs[r])
    acc = g if acc is None else {k: acc[k] + g[k] for k in g}
for k in acc:
    got = lay.view(flat, k).numpy()
    assert np.abs(got - acc[k]).max() <= 1e-12 * (np.abs(acc[k]).max() + 1e-30), k
# every rank then applies the identical clip + Adam -> replicas stay bit-identical
m = {k: np.zeros_like(v) for k, v in p.items()}; v_ = {k: np.zeros_like(v) for k, v in p.items()}
g_all = {k: lay.view(flat, k).numpy().copy() for k in p}
om.clip_and_adam(p, g_all, m, v_, 1, 3e-4, 1.0)
chk = torch.tensor([float(sum(np.abs(v).sum() for v in p.values()))], dtype=torch.float64)
lo, hi = chk.clone(), chk.clone()
dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
assert float(lo) == float(hi)
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_data_parallel_allreduce_equals_gradient_accumulation_gloo(tmp_path):
    """N ranks x batch b == the reference's mini_batch_size = N accumulation (:391-406); world_size 2 on gloo."""
    script = tmp_path / "dp_worker.py"
    script.write_text(_DP_WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29541", str(script)],
                         env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert out.stdout.count("ok") == 2


_DP_MODEL_WORKER = r"""
import os, sys, tempfile, numpy as np, torch
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
from rnn_speech_amd import dataparallel
from dp_oracle_engine import OracleAcousticModel, ListDataset
from oracle import model as om
grp = dataparallel.current()                       # joins from RANK / WORLD_SIZE (gloo: no GPU here)
rank, world = grp.rank, grp.world
L, H, D, C, B, T, U = 1, 8, 5, 80, 2, 10, 4
def batch(seed):
    rng = np.random.RandomState(seed)
    x = rng.randn(T, B, D).astype(np.float32)
    ln = np.array([10, 6 + seed %% 4], np.int32)
    d = np.zeros((B, U), np.int32); d[:, 0] = [3 + seed %% 5, 9]; d[:, 1] = [5, 79]; d[0, 2] = 79
    return x, ln, d
# UNEQUAL shards: rank 0 holds 3 mini-batches, rank 1 only 2
mine = [batch(100 * r + i) for r in range(world) for i in range(3 - r)][0 if rank == 0 else 3:][: 3 - rank]
everyone = {r: [batch(100 * r + i) for i in range(3 - r)] for r in range(world)}
model = OracleAcousticModel(L, H, B, T, U, D, False, C)
model.create_training_rnn(1.0, 1.0, 1.0, 3e-3, 0.5, use_iterator=True)
it, vit = model.add_datasets_input(ListDataset(mine), ListDataset([]))
it.initializer(); vit.initializer()
ref_p = om.init_params(L, H, D, C, seed=0, dtype=np.float64)
ref_p = {k: v.astype(np.float64) for k, v in model.engine.to_numpy().items()}
ref_m = {k: np.zeros_like(v) for k, v in ref_p.items()}; ref_v = {k: np.zeros_like(v) for k, v in ref_p.items()}
ref_step = 0
def ref_apply(batches):                           # the reference's mini_batch_size accumulation over `batches`
    global ref_step
    ref_step += 1
    acc = {k: np.zeros_like(v) for k, v in ref_p.items()}
    losses = []
    for x, ln, d in batches:
        lg, _, cache = om.forward(ref_p, x.astype(np.float64), ln, L, keep_cache=True)
        loss, dl = om.ctc_loss_and_grad(lg, om.sparsify_labels(d, C), ln)
        g = om.backward(ref_p, cache, dl, ln, L)
        for k in acc: acc[k] += g[k]
        losses.append(float(np.mean(loss / ln)))
    om.clip_and_adam(ref_p, acc, ref_m, ref_v, ref_step, 3e-3, 1.0)
    return float(np.mean(losses))
def same_everywhere(vals):
    lo = grp.sum_scalars([v for v in vals]); return [a / world for a in lo]
log = []
# epoch 1: two full steps (both ranks have data), then the epoch ends TOGETHER although rank 0 has a batch left
for step in range(2):
    loss, err, gs, empty = model.run_train_step(None, 1, 1.0)
    want = ref_apply([everyone[r][step] for r in range(world)])
    assert not empty and gs == step + 1 and abs(loss - want) < 1e-5 * abs(want), (loss, want)
    log += [loss, err]
loss, err, gs, empty = model.run_train_step(None, 1, 1.0)
assert empty and gs == 2 and loss == 0.0            # nobody stepped: no rank is left alone in an all-reduce
# epoch 2 with mini_batch_size = 2: the second optimiser step is PARTIAL (one mini-batch) on every rank
it.initializer()
loss, err, gs, empty = model.run_train_step(None, 2, 1.0)
want = ref_apply([everyone[r][i] for i in range(2) for r in range(world)])
assert not empty and gs == 3 and abs(loss - want) < 1e-5 * abs(want), (loss, want)
log += [loss, err]
loss, err, gs, empty = model.run_train_step(None, 2, 1.0)
assert empty and gs == 3, (empty, gs)               # rank 1 is out of data -> nobody runs a mini-batch
# replicas identical to each other and to the single-process accumulation oracle
got = model.engine.to_numpy()
for k in ref_p:
    assert np.abs(got[k] - ref_p[k]).max() < 2e-5, (k, np.abs(got[k] - ref_p[k]).max())
flat = model.engine.params.double()
chk = [float(flat.sum()), float((flat * flat).sum())] + log
mean = same_everywhere(chk)
assert all(abs(a - b) <= 1e-12 * max(1.0, abs(b)) for a, b in zip(chk, mean)), (chk, mean)
# checkpoint: rank 0 writes (with Adam moments), everyone restores the same state
ckpt = grp.broadcast_object(tempfile.mkdtemp() if rank == 0 else None)
model.save(None, ckpt)
assert os.path.exists(os.path.join(ckpt, "checkpoint"))
fresh = OracleAcousticModel(L, H, B, T, U, D, False, C)
fresh.create_training_rnn(1.0, 1.0, 1.0, 1.0, 0.5, use_iterator=True)
fresh.engine.params.add_(float(rank))               # replicas deliberately different before the restore
fresh.restore(None, ckpt)
assert torch.equal(fresh.engine.params, model.engine.params) and torch.equal(fresh.engine.adam_m, model.engine.adam_m)
assert torch.equal(fresh.engine.adam_v, model.engine.adam_v) and fresh.engine.adam_step == 3
assert fresh.global_step.value == 3 and abs(fresh.get_learning_rate() - 3e-3) < 1e-9
grp.barrier()
print("rank", rank, "ok")
"""


def test_data_parallel_drop_in_loop_two_ranks_gloo(tmp_path):
    """AcousticModel.run_train_step on 2 ranks (gloo, oracle-backed engine on CPU tensors) across an epoch boundary
    with UNEQUAL shards: Engine.all_reduce_grads is what exchanges the gradients, no rank enters a collective alone,
    the logged scalars are job-wide means, replicas stay identical and equal the reference's gradient accumulation
    (:391-406, :916-926); rank 0 saves, everyone restores (Adam moments included)."""
    script = tmp_path / "dp_model_worker.py"
    script.write_text(_DP_MODEL_WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29547")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29547", str(script)],
                         env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert out.stdout.count("ok") == 2


_DP_RECOVER_WORKER = r"""
import os, sys, numpy as np, torch
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
from rnn_speech_amd import dataparallel
from dp_oracle_engine import OracleAcousticModel, ListDataset
from oracle import model as om
grp = dataparallel.current()
rank, world = grp.rank, grp.world
L, H, D, C, B, T, U = 1, 8, 5, 80, 2, 10, 4
def batch(seed):
    rng = np.random.RandomState(seed)
    x = rng.randn(T, B, D).astype(np.float32)
    ln = np.array([10, 6 + seed %% 4], np.int32)
    d = np.zeros((B, U), np.int32); d[:, 0] = [3 + seed %% 5, 9]; d[:, 1] = [5, 79]; d[0, 2] = 79
    return x, ln, d
everyone = {r: [batch(100 * r + i) for i in range(6)] for r in range(world)}
model = OracleAcousticModel(L, H, B, T, U, D, False, C)
model.create_training_rnn(1.0, 1.0, 1.0, 3e-3, 0.5, use_iterator=True)
it, vit = model.add_datasets_input(ListDataset(everyone[rank]), ListDataset([]))
it.initializer(); vit.initializer()
ref_p = {k: v.astype(np.float64) for k, v in model.engine.to_numpy().items()}
ref_m = {k: np.zeros_like(v) for k, v in ref_p.items()}; ref_v = {k: np.zeros_like(v) for k, v in ref_p.items()}
ref_step = 0
def ref_apply(batches):
    global ref_step
    ref_step += 1
    acc = {k: np.zeros_like(v) for k, v in ref_p.items()}
    for x, ln, d in batches:
        lg, _, cache = om.forward(ref_p, x.astype(np.float64), ln, L, keep_cache=True)
        loss, dl = om.ctc_loss_and_grad(lg, om.sparsify_labels(d, C), ln)
        g = om.backward(ref_p, cache, dl, ln, L)
        for k in acc: acc[k] += g[k]
    om.clip_and_adam(ref_p, acc, ref_m, ref_v, ref_step, 3e-3, 1.0)
def agree_params():
    got = model.engine.to_numpy()
    for k in ref_p:
        assert np.abs(got[k] - ref_p[k]).max() < 2e-5, (k, np.abs(got[k] - ref_p[k]).max())
    flat = model.engine.params.double()
    chk = grp.sum_scalars([float(flat.sum()) * (1 if rank == 0 else -1), float((flat * flat).sum()) * (1 if rank == 0 else -1)])
    assert all(abs(v) < 1e-12 for v in chk), chk
eng = model.engine
# step 1 (mini_batch_size 2): the SECOND mini-batch times out on rank 1 only -> its contribution is taken back (the first one's is
# kept), the mini-batch is repeated per diagonal, and the step is what the reference's accumulation over all four mini-batches gives
loss, err, gs, empty = model.run_train_step(None, 1, 1.0)
ref_apply([everyone[r][0] for r in range(world)]); agree_params()
class FailSecond(object):                         # rank 1: health checks 1 and 3 of this step pass, 2 fails
    pass
calls = {"n": 0}
orig_check = eng.check
def check_second_fails():
    calls["n"] += 1
    if rank == 1 and calls["n"] == 2:
        from rnn_speech_amd.lib import DataflowTimeout
        raise DataflowTimeout("injected time-out")
eng.check = check_second_fails
loss, err, gs, empty = model.run_train_step(None, 2, 1.0)
eng.check = orig_check
assert gs == 2 and not empty
assert model.recovered_steps == (1 if rank == 1 else 0) and getattr(eng, "per_diagonal_runs", 0) == (1 if rank == 1 else 0)
ref_apply([everyone[r][i] for i in (1, 2) for r in range(world)]); agree_params()
# step 2: rank 0's mini-batch is invalid twice (the repeat fails too): NO rank applies the step, nobody hangs in the all-reduce
if rank == 0:
    eng.fail_checks = 2
loss, err, gs, empty = model.run_train_step(None, 1, 1.0)
assert gs == 2 and not empty, (gs, empty)         # the step counter did not move, on either rank
assert model.skipped_steps == 1 and model.recovered_steps == (1 if rank == 1 else 1)
agree_params()                                    # parameters untouched and identical
# step 3: business as usual (the dropped step's gradients did not linger)
loss, err, gs, empty = model.run_train_step(None, 1, 1.0)
assert gs == 3
ref_apply([everyone[r][4] for r in range(world)]); agree_params()
# step 4 (ADVICE r5): batch norm over the GLOBAL batch -- forward and backward of a mini-batch are cross-rank collectives, so a rank must
# NOT repeat a mini-batch alone (the repeat's collectives would pair with the peers' next ones): no per-diagonal run, the step is
# dropped on every rank, the invalid mini-batch adds nothing to the logged loss, parameters stay identical
eng.sync_batch_norm = True
runs_before, skipped_before = getattr(eng, "per_diagonal_runs", 0), model.skipped_steps
if rank == 1:
    eng.fail_checks = 1
loss, err, gs, empty = model.run_train_step(None, 1, 1.0)
eng.sync_batch_norm = False
assert getattr(eng, "per_diagonal_runs", 0) == runs_before, "a rank repeated a mini-batch alone under sync_batch_norm"
assert gs == 3 and model.skipped_steps == skipped_before + 1
assert np.isfinite(loss)
agree_params()
grp.barrier()
print("rank", rank, "ok")
"""


def test_data_parallel_time_out_recovery_two_ranks_gloo(tmp_path):
    """a dataflow time-out no longer ends training.  run_step takes the invalid mini-batch's gradient contribution
    back, repeats it on the launch-per-diagonal kernels (Engine.mini_batch(per_diagonal=True)) and goes on -- the optimiser step
    equals the reference's accumulation (:391-406, :887-939); when the repeat is invalid too the ranks agree over the host group
    BEFORE the all-reduce and every rank drops the step.  2 ranks on gloo, oracle-backed engine, injected failures."""
    script = tmp_path / "dp_recover_worker.py"
    script.write_text(_DP_RECOVER_WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29557")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29557", str(script)],
                         env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert out.stdout.count("ok") == 2


def test_lstm_workspace_covers_every_shorter_run_length():
    """Engine sizes ONE allocation for max_T and re-lays it out per mini-batch (ops.LstmWorkspace.prefix), so
    amdspeech_lstm_workspace_bytes(T) must cover the layout of every T' <= T.  It did not where a region exists only for some T:
    the bf16 operand copies of precision = bf16 at 1024 units (only when T * B is a multiple of 64) and the regions behind the
    32-bit buffer resources of the whole-sequence kernels (only below a sequence length).  A host function: no GPU needed."""
    import ctypes
    from rnn_speech_amd import lib as _l
    lib = _l.load()

    def nbytes(T, B, H, L, precision):
        d = _l.LstmDesc(T, B, H, L, 1.0, 1.0, 0, precision)
        n = lib.amdspeech_lstm_workspace_bytes(ctypes.byref(d))
        assert n > 0, (T, B, H, L, precision)
        return n

    for (T, B, H, L, pr) in [(1001, 32, 1024, 1, 2), (1001, 32, 1024, 3, 2), (1001, 16, 1024, 5, 2), (1001, 48, 1024, 5, 2),
                             (998, 64, 1024, 5, 2), (1001, 32, 512, 3, 0), (6000, 32, 512, 3, 0), (5470, 32, 512, 3, 0),
                             (17000, 32, 512, 3, 0), (2100, 16, 512, 8, 0), (1001, 32, 256, 3, 1)]:
        root = nbytes(T, B, H, L, pr)
        step = 1 if T <= 1100 else 37
        worst = max(nbytes(t, B, H, L, pr) for t in list(range(1, T, step)) + [T - 1])
        assert worst <= root, (T, B, H, L, pr, worst, root)


_DP_BUCKET_WORKER = r"""
import os, sys, numpy as np, torch
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
from rnn_speech_amd import dataparallel
from dp_oracle_engine import OracleAcousticModel, ListDataset
grp = dataparallel.current()
rank, world = grp.rank, grp.world
L, H, D, C, B, T, U = 1, 8, 5, 80, 3, 24, 3
rng = np.random.RandomState(5)
N = 41                                              # 41 utterances: 6 global buckets of B * world = 6, and a tail of 5 (padded to 6)
dur = np.round(rng.uniform(0.4, 2.4, size=N), 3)    # seconds; 10 frames per second below
items = [["utt%%02d" %% i, "label", float(dur[i])] for i in range(N)]
seed = grp.broadcast_object(1234 if rank == 0 else None)
mine = dataparallel.shard_bucketed(items, B, rank, world, seed)
# every rank holds the same number of utterances (equal number of collectives), and a rank-local draw would not
counts = grp.sum_scalars([len(mine) if r == rank else 0 for r in range(world)])
assert len(set(counts)) == 1 and counts[0] == 21, counts
ordered = sorted(items, key=lambda it: it[2])
per = B * world
buckets = [ordered[i:i + per] for i in range(0, N, per)]
def bucket_of(name):
    return [k for k, g in enumerate(buckets) if any(it[0] == name for it in g)][0]
def batch(chunk, k):
    n = len(chunk)
    ln = np.array([max(2, int(round(it[2] * 10))) for it in chunk] + [0] * (B - n), np.int32)
    x = np.random.RandomState(100 + k).randn(T, B, D).astype(np.float32)
    d = np.zeros((B, U), np.int32); d[:n, 0] = 3 + k %% 5; d[:n, 1] = 79
    return x, ln, d
chunks = [mine[i:i + B] for i in range(0, len(mine), B)]
model = OracleAcousticModel(L, H, B, T, U, D, False, C)
model.create_training_rnn(1.0, 1.0, 1.0, 3e-3, 0.5, use_iterator=True)
it, vit = model.add_datasets_input(ListDataset([batch(c, k) for k, c in enumerate(chunks)]), ListDataset([]))
it.initializer(); vit.initializer()
seen = []
for k, chunk in enumerate(chunks):
    loss, err, gs, empty = model.run_train_step(None, 1, 1.0)        # the PRODUCT's step: one all-reduce per call
    assert not empty and gs == k + 1
    bk = bucket_of(chunk[0][0])
    assert all(bucket_of(it_[0]) == bk for it_ in chunk)             # a mini-batch never straddles two global buckets
    longest = max(it_[2] for it_ in chunk)
    both = grp.sum_scalars([longest if r == rank else 0.0 for r in range(world)] + [float(bk) if rank == 0 else 0.0, float(bk) if rank == 1 else 0.0])
    assert both[world] == both[world + 1], both                      # step k is the SAME global bucket on every rank
    width = buckets[bk][-1][2] - buckets[bk][0][2]
    assert abs(both[0] - both[1]) < max(width, 1e-9), (k, both, width)     # ... and the ranks' longest utterances are neighbours in it
    seen.append(bk)
assert sorted(seen) == list(range(len(buckets))) and seen[-1] == len(buckets) - 1      # every bucket once, the short one last
assert seen[:-1] != sorted(seen[:-1])                                # ... in the job's shuffled order
loss, err, gs, empty = model.run_train_step(None, 1, 1.0)
assert empty                                                         # the epoch ends on every rank together
# the end-of-epoch re-draw keeps the buckets together and is the same permutation everywhere
again = dataparallel.reshuffle_buckets(mine, B, grp.broadcast_object(99 if rank == 0 else None))
order = [bucket_of(again[i][0]) for i in range(0, len(again), B)]
tot = grp.sum_scalars([float(v) * (1 if rank == 0 else -1) for v in order])
assert all(t == 0.0 for t in tot) and sorted(order) == list(range(len(buckets)))
flat = model.engine.params.double()
chk = grp.sum_scalars([float(flat.sum()) * (1 if rank == 0 else -1)])
assert abs(chk[0]) < 1e-9                                            # replicas identical
grp.barrier()
print("rank", rank, "ok")
"""


def test_data_parallel_global_length_buckets_two_ranks_gloo(tmp_path):
    """SURVEY 8e "with bucketing, shard within a bucket so ranks see similar T": dataparallel.shard_bucketed cuts the job's
    length-sorted utterances into global buckets of batch_size * world, deals each bucket across the ranks and shuffles the
    buckets with the job's seed.  Driven through the product's run_train_step on 2 ranks (gloo, oracle-backed engine): every
    optimiser step is one global bucket on both ranks, the ranks' longest utterances differ by less than the bucket's width,
    shards are equal, the epoch ends together, the re-draw at the end of an epoch is the same permutation on every rank."""
    script = tmp_path / "dp_bucket_worker.py"
    script.write_text(_DP_BUCKET_WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29553")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29553", str(script)],
                         env=env, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert out.stdout.count("ok") == 2


_BOOTSTRAP_WORKER = r"""
import os, sys, torch
sys.path.insert(0, %(root)r)
import torch.distributed as dist
from rnn_speech_amd import dataparallel
from rnn_speech_amd.lib import AmdSpeechError
mode = sys.argv[1]
os.environ["AMDSPEECH_COMM"] = "rccl" if mode == "asym" else mode
if mode == "asym":
    # ONE rank cannot bind RCCL (ADVICE r3): nobody may enter ncclCommInitRank -- a collective -- or the able ranks hang in it
    from rnn_speech_amd import lib as _l
    lib = _l.load()
    able = os.environ["RANK"] == "0"
    lib.amdspeech_comm_unique_id = lambda buf: 0 if able else -3
    lib.amdspeech_comm_available = lambda: 0 if able else -3
    def _never(*a):
        print("comm_init ENTERED on rank", os.environ["RANK"])
        return -1
    lib.amdspeech_comm_init = _never
    mode = "rccl"
# no GPU in this container: ncclCommInitRank (or already ncclGetUniqueId) fails on every rank -- the bootstrap must come out
# of it on ALL ranks together: rank 0 always broadcasts (an id or None), nobody is left alone in a collective
try:
    grp = dataparallel.Group.from_env()
    assert mode == "auto" or mode == "rccl-soft", mode
    assert grp.device_channel == "torch-gloo" and grp.comm_info() is None
    t = torch.ones(4) * (grp.rank + 1)
    grp.all_reduce_sum_(t)
    assert float(t[0]) == 3.0
    print("rank", grp.rank, "fallback ok")
except AmdSpeechError as exc:
    assert mode == "rccl" and "forbids" in str(exc), (mode, exc)
    print("rank", os.environ["RANK"], "strict ok")
dist.destroy_process_group()
"""


@pytest.mark.parametrize("mode", ["rccl", "auto", "asym"])
def test_rccl_bootstrap_failure_is_agreed_by_all_ranks(tmp_path, mode):
    """a failing amdspeech_comm_unique_id / comm_init must not leave peers blocked.  AMDSPEECH_COMM=rccl (what
    bench.py --gpus N sets) turns the missing communicator into an error on EVERY rank; auto (gloo backend) never tries."""
    script = tmp_path / "bootstrap_worker.py"
    script.write_text(_BOOTSTRAP_WORKER % {"root": ROOT})
    port = {"rccl": "29561", "auto": "29563", "asym": "29565"}[mode]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", port, str(script), mode],
                         env=env, capture_output=True, text=True, timeout=200)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert out.stdout.count("fallback ok" if mode == "auto" else "strict ok") == 2
    assert "comm_init ENTERED" not in out.stdout


def test_shard_gives_every_rank_the_same_number_of_items():
    from rnn_speech_amd import dataparallel as dp
    items = list(range(11))
    shards = [dp.shard(items, r, 4) for r in range(4)]
    assert all(len(s) == 3 for s in shards)
    assert sorted(set(sum(shards, []))) == items                 # everything is seen; one item is repeated to pad
    assert dp.shard(items, 0, 1) == items and dp.shard([], 1, 2) == []


def test_stt_cli_surface():
    """The reference's flags (stt.py:360-404) parse; README's stale --train does not exist there either."""
    import importlib
    sys.argv = ["stt.py", "--train_acoustic", "--config", "x.ini", "--max_epoch", "3", "--learn_rate", "0.001"]
    stt = importlib.import_module("stt")
    p = stt.parse_args()
    assert p["train_acoustic"] and p["config_file"] == "x.ini" and p["max_epoch"] == 3 and p["learn_rate"] == 0.001
    sys.argv = ["stt.py", "--file", "a.wav"]
    assert stt.parse_args()["file"] == "a.wav"


def _brute_force_best_labelling(lg, blank):
    import itertools
    T, C = lg.shape
    lp = lg - np.log(np.exp(lg).sum(1, keepdims=True))
    tot = {}
    for path in itertools.product(range(C), repeat=T):
        s = sum(lp[t, k] for t, k in enumerate(path))
        out, prev = [], -1
        for k in path:
            if k != prev and k != blank:
                out.append(k)
            prev = k
        tot[tuple(out)] = np.logaddexp(tot.get(tuple(out), -np.inf), s)
    best = max(tot.items(), key=lambda kv: kv[1])
    return list(best[0]), best[1]


def test_host_beam_search_is_exact_with_a_wide_beam():
    """prefix beam search (stands where tf.nn.ctc_beam_search_decoder sits at :312).  With a
    beam wider than the number of prefixes it must return the most probable labelling (brute force)."""
    from rnn_speech_amd import ops
    rng = np.random.RandomState(0)
    for _ in range(25):
        T, C = rng.randint(1, 7), rng.randint(2, 5)
        lg = (rng.randn(T, 1, C) * 2).astype(np.float32)
        ids, n, lp = ops.ctc_beam_search(lg, [T], beam_width=1000, merge_repeated=False)
        ref, score = _brute_force_best_labelling(lg[:, 0, :].astype(np.float64), C - 1)
        assert list(ids[0, :n[0]]) == ref
        assert abs(lp[0] - score) < 1e-4
        assert np.all(ids[0, n[0]:] == C)


def test_beam_search_merge_repeated_and_lengths():
    from rnn_speech_amd import ops
    from rnn_speech_amd.acoustic_model import _merge_repeated
    lg = np.full((6, 2, 4), -10.0, np.float32)
    for t, k in enumerate([0, 1, 1, 3, 1, 1]):       # "a b b * b b" -> a b b ; merge_repeated -> a b
        lg[t, :, k] = 10.0
    ids, n, _ = ops.ctc_beam_search(lg, [6, 3], beam_width=100, merge_repeated=False)
    assert list(ids[0, :n[0]]) == [0, 1, 1] and list(ids[1, :n[1]]) == [0, 1]
    ids, n, _ = ops.ctc_beam_search(lg, [6, 0], beam_width=100, merge_repeated=True)
    assert list(ids[0, :n[0]]) == [0, 1] and n[1] == 0
    m, ln = _merge_repeated(np.array([[0, 1, 1, 2, 4, 4]]), np.array([4]), 4)
    assert list(m[0, :ln[0]]) == [0, 1, 2]


def test_tf_bundle_reader_on_the_reference_checkpoint_index(golden_dir):
    """the pre-trained model's TensorFlow bundle index (a data file shipped with
    the reference; its 101 MB data blob is a git-LFS pointer) parses to the 3x1024 / 120-dim layout."""
    from rnn_speech_amd import tf_bundle
    path = os.path.join(golden_dir, "reference_acousticmodel.ckpt.index")
    e = tf_bundle.read_index(path)
    assert e[""]["num_shards"] == 1
    assert e["Input_Layer/input_w"]["shape"] == [120, 1024] and e["Input_Layer/input_w"]["offset"] == 4096
    assert e["Output_layer/output_w"]["shape"] == [1024, 80]
    assert e["global_step"]["dtype"] == 3 and e["global_step"]["shape"] == []
    for l, off in enumerate((840008, 34410824, 67981640)):
        k = e["rnn/multi_rnn_cell/cell_%d/basic_lstm_cell/kernel" % l]
        assert k["shape"] == [2048, 4096] and k["offset"] == off and k["size"] == 33554432
    assert max(v["offset"] + v["size"] for n, v in e.items() if n) == 101536072      # = LFS pointer size
    assert tf_bundle.crc32c(b"123456789") == 0xE3069283                               # CRC-32C check value
    assert tf_bundle.verify_index_checksums(path)                                      # TensorFlow's own block CRCs


def test_tf_bundle_write_read_round_trip(tmp_path):
    from rnn_speech_amd import tf_bundle
    rng = np.random.RandomState(0)
    t = {"Input_Layer/input_w": rng.randn(12, 16).astype(np.float32),
         "Input_Layer/input_b": rng.randn(16).astype(np.float32),
         "Output_layer/output_w": rng.randn(16, 80).astype(np.float32),
         "global_step": np.int32(77), "learning_rate": np.float32(3e-4),
         "rnn/multi_rnn_cell/cell_0/basic_lstm_cell/kernel": rng.randn(32, 64).astype(np.float32)}
    prefix = str(tmp_path / "acousticmodel.ckpt-77")
    tf_bundle.write_bundle(prefix, t)
    assert tf_bundle.verify_index_checksums(prefix + ".index")
    idx = tf_bundle.read_index(prefix + ".index")
    assert idx["global_step"]["shape"] == [] and idx["Input_Layer/input_w"]["shape"] == [12, 16]
    back = tf_bundle.read_bundle(prefix)
    assert sorted(back) == sorted(t)
    for k in t:
        assert back[k].dtype == np.asarray(t[k]).dtype and np.array_equal(back[k], np.asarray(t[k]))
    # per-tensor checksum is the masked CRC32C of the raw bytes (what tf.train.Saver verifies on restore)
    raw = np.asarray(t["Input_Layer/input_b"]).tobytes()
    assert idx["Input_Layer/input_b"]["crc32c"] == tf_bundle._mask(tf_bundle.crc32c(raw))
    # a flipped bit in the data shard is caught by read_bundle (it used to load silently)
    shard = prefix + ".data-00000-of-00001"
    blob = bytearray(open(shard, "rb").read())
    blob[idx["Output_layer/output_w"]["offset"] + 5] ^= 0x10
    open(shard, "wb").write(bytes(blob))
    with pytest.raises(IOError, match="crc32c"):
        tf_bundle.read_bundle(prefix)


# ----------------------------------------------------------------------------- corpus discovery
from corpus_fixture import build_trees, write_wav as _write_wav      # noqa: E402


def test_corpus_walkers_match_reference_golden(tmp_path, golden_dir):
    """tests/golden/corpus_walk.json = the reference's own DataProcessor (tools/make_golden.py) over the
    synthetic trees of tests/corpus_fixture.py: type probe (:208-226), the four layouts (:263-329), label
    cleaning and the text / duration filters (:64-67)."""
    from util.dataprocessor import DataProcessor
    gold = json.load(open(os.path.join(golden_dir, "corpus_walk.json")))
    dirs, _ = build_trees(tmp_path, ted_segments=True)
    for d in dirs:
        assert DataProcessor.get_type(d) == gold["types"][os.path.basename(d)]
    for d in dirs + [",".join(dirs)]:
        key = ",".join(os.path.basename(x) for x in d.split(","))
        got = sorted([os.path.relpath(os.path.normpath(a), str(tmp_path)), t, round(float(n), 6)]
                     for a, t, n in DataProcessor(d).get_dataset())
        assert got == gold["datasets"][key], key
    assert DataProcessor.get_type(str(tmp_path / "libri" / "19" / "198" / "19-198-0000.flac")) == "Unrecognized"
    with pytest.raises(Exception):
        DataProcessor(str(tmp_path / "nothing_here"))


def test_corpus_cache_and_native_sphere_segments(tmp_path):
    """The interchangeable file-list cache (:251-261) and the TED-LIUM segments cut natively from the
    .sph (the reference shells out to sox, :331-337) -- same file names, exact samples."""
    import pickle
    import wave
    from util.dataprocessor import DataProcessor
    dirs, ramp = build_trees(tmp_path, ted_segments=False)
    libri, ted = dirs[0], dirs[3]
    cache = str(tmp_path / "cache.pkl")
    assert len(DataProcessor(libri, file_cache=cache).get_dataset()) == 4
    paths, cached = pickle.load(open(cache, "rb"))                     # the reference's cache layout
    assert paths == [libri] and len(cached) == 5                      # the unfiltered list is what is cached
    os.remove(os.path.join(libri, "19", "198", "19-198-0001.flac"))   # the cache, not the tree, is now the source
    assert len(DataProcessor(libri, file_cache=cache).get_dataset()) == 4
    assert len(DataProcessor(libri + ", " + libri, file_cache=cache).get_dataset()) == 6   # other dirs: rescan
    got = DataProcessor(ted).get_dataset()
    assert [(os.path.basename(a), t, round(d, 3)) for a, t, d in got] == [
        ("TalkA_0.5.wav", "the first segment of speech", 1.5), ("TalkA_3.0.wav", "and the last one", 0.9)]
    with wave.open(got[0][0], "rb") as w:
        seg = np.frombuffer(w.readframes(w.getnframes()), "<i2")
    assert np.array_equal(seg, ramp[8000:32000])                       # big-endian source, exact samples


def test_load_acoustic_dataset_split_and_manifest(tmp_path):
    """SpeechRecognizer.load_acoustic_dataset (reference models/SpeechRecognizer.py:58-99)."""
    from models.SpeechRecognizer import SpeechRecognizer
    d = tmp_path / "vy"
    d.mkdir()
    for i in range(10):
        _write_wav(str(d / ("u%d.wav" % i)), 0.5 + 0.1 * i)
        (d / ("u%d.wav.trn" % i)).write_text("utterance number %d\n" % i)
    train, test = SpeechRecognizer.load_acoustic_dataset(str(d), None, None, ordered=True, train_frac=0.8)
    assert [round(x[2], 1) for x in train] == [0.5, 0.6, 0.7, 0.8, 0.9, 1.0, 1.1, 1.2] and len(test) == 2
    train, test = SpeechRecognizer.load_acoustic_dataset(str(d), str(d))
    assert len(train) == 10 and len(test) == 10
    train, test = SpeechRecognizer.load_acoustic_dataset(str(d))
    assert len(train) == 10 and test == []
    tsv = tmp_path / "m.tsv"
    tsv.write_text("%s\tHello World\n" % (d / "u3.wav"))
    train, _ = SpeechRecognizer.load_acoustic_dataset(str(tsv))
    assert train == [[str(d / "u3.wav"), "hello world", 0.8]]


def _dict_prefix_beam_search(lg, blank, width):
    """The prefix beam search restated with a dictionary keyed by whole prefixes (slow and obviously right about WHICH
    candidates are the same prefix): returns (best prefix, its log probability)."""
    T, C = lg.shape
    lp = (lg - np.log(np.exp(lg.astype(np.float64)).sum(1, keepdims=True))).astype(np.float32)
    NEG = -np.inf

    def lse(a, b):
        return np.float32(np.logaddexp(a, b)) if (a != NEG or b != NEG) else NEG

    beams = {(): (np.float32(0.0), NEG)}                  # prefix -> (p_blank, p_non_blank)
    for t in range(T):
        nxt = {}
        for pre, (pb, pnb) in sorted(beams.items()):
            tot = lse(pb, pnb)
            b0, n0 = nxt.get(pre, (NEG, NEG))
            b0 = lse(b0, tot + lp[t, blank])
            if pre:
                n0 = lse(n0, pnb + lp[t, pre[-1]])
            nxt[pre] = (b0, n0)
            for c in range(C):
                if c == blank:
                    continue
                frm = pb if (pre and pre[-1] == c) else tot
                if frm == NEG:
                    continue
                b1, n1 = nxt.get(pre + (c,), (NEG, NEG))
                nxt[pre + (c,)] = (b1, lse(n1, frm + lp[t, c]))
        order = sorted(nxt.items(), key=lambda kv: (-lse(*kv[1]), kv[0]))
        beams = dict(order[:width])
    best = min(beams.items(), key=lambda kv: (-lse(*kv[1]), kv[0]))
    return list(best[0]), float(lse(*best[1]))


def test_host_beam_search_narrow_beam_merges_every_duplicate_prefix():
    """A small alphabet, long inputs, a beam narrower than the number of live prefixes: prefixes fall out of the beam and come
    back while their extensions are still in it -- every such meeting has to MERGE (one entry per prefix), or probability mass is
    split over duplicates.  Against the dictionary restatement, prefix and log probability."""
    from rnn_speech_amd import ops
    rng = np.random.RandomState(3)
    for trial in range(12):
        T, C, width = rng.randint(20, 60), rng.randint(3, 6), int(rng.choice([3, 8, 25]))
        lg = (rng.randn(T, 1, C) * rng.choice([0.5, 2.0])).astype(np.float32)
        ids, n, lp = ops.ctc_beam_search(lg, [T], beam_width=width, merge_repeated=False)
        ref, score = _dict_prefix_beam_search(lg[:, 0, :], C - 1, width)
        assert list(ids[0, :n[0]]) == ref, trial
        assert abs(lp[0] - score) < 1e-3, (trial, lp[0], score)


def test_bounded_beam_selection_is_bit_identical_to_the_exhaustive_scan():
    """Round 3's decoder scores only the (entry, label) pairs under the hyperbola (rank_entry + 1) * rank_label <= width (the rest
    cannot be in the beam but through an exact tie, which falls back) -- against the exhaustive scan of round 2
    (AMDSPEECH_BEAM_EXHAUSTIVE=1, read per call): prefixes, lengths and log probabilities bit for bit, on random posteriors of
    every sharpness, blank-heavy ones, and inputs made of exact ties (rounded / all-equal logits)."""
    from rnn_speech_amd import ops
    rng = np.random.RandomState(0)

    def run(lg, lens, w, merge, exhaustive):
        os.environ["AMDSPEECH_BEAM_EXHAUSTIVE"] = "1" if exhaustive else "0"
        try:
            return ops.ctc_beam_search(lg, lens, beam_width=w, merge_repeated=bool(merge))
        finally:
            os.environ.pop("AMDSPEECH_BEAM_EXHAUSTIVE", None)
    cases = 0
    for trial in range(60):
        T, B, C = rng.randint(5, 70), rng.randint(1, 5), int(rng.choice([3, 5, 20, 80]))
        lg = (rng.randn(T, B, C) * rng.choice([0.5, 2.0, 6.0])).astype(np.float32)
        if trial % 3 == 0:
            lg[:, :, C - 1] += 4.0
        if trial % 7 == 0:
            lg = np.round(lg)                                   # exact ties
        if trial % 11 == 0:
            lg[:] = 0.0                                         # nothing but ties
        if trial % 13 == 0:
            lg = np.round(lg * 2) / 2
        lens = rng.randint(0, T + 1, size=B).astype(np.int32)
        for w in (1, 3, 25, 100):
            a, b = run(lg, lens, w, trial % 2, True), run(lg, lens, w, trial % 2, False)
            cases += 1
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]), (trial, w)
    assert cases == 240


def test_beam_search_thread_cap_changes_nothing_but_the_thread_count():
    """amdspeech_ctc_beam_search_host_mt (the asynchronous training-time decoder's entry): any cap on the decode threads gives the
    results of the uncapped call, bit for bit."""
    from rnn_speech_amd import ops
    rng = np.random.RandomState(11)
    lg = rng.randn(50, 7, 30).astype(np.float32)
    lens = rng.randint(0, 51, size=7).astype(np.int32)
    ref = ops.ctc_beam_search(lg, lens, beam_width=20)
    for cap in (1, 2, 3, 64):
        out = ops.ctc_beam_search(lg, lens, beam_width=20, max_threads=cap)
        assert all(np.array_equal(u, v) for u, v in zip(ref, out)), cap


def test_bounded_beam_selection_at_large_widths_and_label_counts():
    """The same equivalence where the round-4 fast paths change over: more than 512 sort keys (the AVX2 rank sort hands back to
    std::sort), beams wider than the label set, two labels, flat / sharp / all-tie posteriors, a zero-length utterance."""
    from rnn_speech_amd import ops
    rng = np.random.RandomState(3)
    for (T, B, C, w) in [(40, 2, 600, 300), (30, 2, 700, 600), (60, 3, 130, 120), (40, 2, 9, 600), (30, 2, 2, 5), (60, 2, 40, 513)]:
        for scale in (0.05, 1.0, 5.0):
            lg = (rng.randn(T, B, C) * scale).astype(np.float32)
            if scale == 5.0:
                lg = np.round(lg)
            lens = np.array([T] + [rng.randint(0, T + 1) for _ in range(B - 2)] + [0], np.int32)[:B]
            out = []
            for exhaustive in ("1", "0"):
                os.environ["AMDSPEECH_BEAM_EXHAUSTIVE"] = exhaustive
                try:
                    out.append(ops.ctc_beam_search(lg, lens, beam_width=w, merge_repeated=True))
                finally:
                    os.environ.pop("AMDSPEECH_BEAM_EXHAUSTIVE", None)
            assert all(np.array_equal(u, v) for u, v in zip(*out)), (T, B, C, w, scale)


def test_host_edit_distance_matches_oracle():
    from rnn_speech_amd import ops
    from oracle import model as om
    rng = np.random.RandomState(1)
    n, la, lb = 40, 23, 17
    a = rng.randint(0, 6, size=(n, la)).astype(np.int32)
    b = rng.randint(0, 6, size=(n, lb)).astype(np.int32)
    al = rng.randint(0, la + 1, size=n).astype(np.int32)
    bl = rng.randint(0, lb + 1, size=n).astype(np.int32)
    got = ops.edit_distance_host(a, al, b, bl)
    want = [om.edit_distance(a[i, :al[i]], b[i, :bl[i]]) for i in range(n)]
    assert list(got) == want


def test_host_beam_search_width_100_is_fast():
    """The reference's decoder setting (width 100) on 4 x 300 frames x 80 labels: well under a second on any host (the first
    implementation, a std::map keyed by prefix vectors, took 4 s here and 45 s for a batch of 32 x 1001 frames)."""
    import time
    from rnn_speech_amd import ops
    rng = np.random.RandomState(0)
    lg = (rng.randn(300, 4, 80) * 3).astype(np.float32)
    t0 = time.time()
    ops.ctc_beam_search(lg, [300] * 4, beam_width=100, merge_repeated=True)
    assert time.time() - t0 < 1.5


def test_bench_gpus_n_invoked_plainly_becomes_its_own_launcher(monkeypatch):
    """`python bench.py --gpus N` with no WORLD_SIZE re-executes itself under torch.distributed.run (one process per GPU,
    127.0.0.1 rendezvous, same arguments) instead of exiting; inside a job whose size disagrees it still refuses."""
    import importlib.util
    import subprocess
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}
    monkeypatch.setattr(subprocess, "call", lambda cmd, **kw: seen.setdefault("cmd", cmd) and 0)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7", "--warmup", "2"])
    with pytest.raises(SystemExit) as ex:
        bench.main()
    assert ex.value.code == 0
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nnodes=1" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"] and cmd[-7].endswith("bench.py")
    monkeypatch.setenv("WORLD_SIZE", "2")
    with pytest.raises(SystemExit) as ex:
        bench.main()
    assert "inside a job of 2 ranks" in str(ex.value.code)


def test_train_decoder_keys_of_config_ini(tmp_path):
    """`train_decoder` / `train_decoder_lag` are config.ini keys (they were attributes only): defaults = the
    reference's beam decoder, one mini-batch late; an unknown decoder name is an error, not a silent fallback."""
    from rnn_speech_amd import hyperparams
    hp = hyperparams.read_config_file(os.path.join(ROOT, "config.ini"))
    assert hp["train_decoder"] == "beam" and hp["train_decoder_lag"] == 1
    txt = open(os.path.join(ROOT, "config.ini")).read().replace("train_decoder : beam", "train_decoder : greedy").replace(
        "train_decoder_lag : 1", "train_decoder_lag : 0")
    p = tmp_path / "c.ini"
    p.write_text(txt)
    hp = hyperparams.read_config_file(str(p))
    assert hp["train_decoder"] == "greedy" and hp["train_decoder_lag"] == 0
    p.write_text(txt.replace("train_decoder : greedy", "train_decoder : viterbi"))
    with pytest.raises(ValueError):
        hyperparams.read_config_file(str(p))
