#!/usr/bin/env python3
"""Headline benchmark: audio-frames/sec of the MFCC -> 3x512 LSTM -> CTC training step.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path over one synthetic mini-batch per rank, PCM already
resident in HBM: MFCC front end (40 coefficients) -> input Linear -> 3x512 LSTM forward ->
output Linear -> CTC loss+grad -> BPTT -> [RCCL all-reduce of the flat gradient] ->
clip_by_global_norm + Adam.  Dropout keep (0.8, 0.5) as the reference trains.  Weak scaling:
every rank processes its own batch of 32 ten-second utterances (BASELINE.json configs[1]).
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

L, H, D, C, B, T, U = 3, 512, 40, 80, 32, 1001, 161      # configs[1] (the headline); --config cfg3 overrides
MODE = "mfcc"
SR, SECONDS = 16000, 10
N_ROTATE = 4                       # distinct PCM / label batches rotated through the timed loop
CONFIGS = {
    # BASELINE.json configs[1]: the configuration the metric is quoted on
    "cfg2": dict(L=3, H=512, D=40, B=32, T=1001, mode="mfcc",
                 metric="audio_frames_per_sec_train_3x512_lstm_ctc",
                 name="configs[1]: 3x512 LSTM + CTC training step, 40-dim MFCC"),
    # BASELINE.json configs[2]: 5x1024, 120-dim mel-filterbank + delta + delta-delta, batch 64
    "cfg3": dict(L=5, H=1024, D=120, B=64, T=998, mode="fbank",
                 metric="audio_frames_per_sec_train_5x1024_lstm_ctc",
                 name="configs[2]: 5x1024 LSTM + CTC training step, 120-dim fbank+delta+delta-delta"),
    # BASELINE.json configs[4], the per-GPU share: 5x1024 BIDIRECTIONAL (two stacks, outputs concatenated), same features and
    # batch as cfg3; exact f32 unless --precision bf16x3 (which at H = 1024 still runs the launch-per-diagonal kernels)
    "cfg5": dict(L=5, H=1024, D=120, B=64, T=998, mode="fbank", bidirectional=True,
                 metric="audio_frames_per_sec_train_5x1024_bidirectional_lstm_ctc",
                 name="configs[4] per GPU: 5x1024 bidirectional LSTM + CTC training step, 120-dim fbank+delta+delta-delta"),
}
FP32_MFMA_PEAK_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md (dense f32-input MFMA)


def synth_pcm(seed, n):
    rng = np.random.RandomState(seed)
    t = np.arange(n) / float(SR)
    sig = 0.1 * rng.randn(n)
    for f0, a in ((220.0, 0.3), (1330.0, 0.2), (3100.0, 0.1)):
        sig += a * np.sin(2 * np.pi * f0 * (1 + 0.01 * (seed % 17)) * t)
    return sig.astype(np.float32)


def synth_labels(rng, batch):
    dense = np.zeros((batch, U), np.int32)
    for b in range(batch):
        n = rng.randint(80, 161)                       # U ~ U[80,160], EOS appended
        dense[b, :n - 1] = rng.randint(1, C - 1, size=n - 1)
        dense[b, n - 1] = C - 1
    return dense


def cpu_baseline(t_sample=None, steps=2):
    """The numpy oracle (a PORT of the reference graph; TensorFlow is not installable here)
    timed on this host: one optimiser step, B=32, on the first `t_sample` frames."""
    from oracle import model as om
    t_sample = T if t_sample is None else t_sample
    # OpenBLAS oversubscribes badly on these small matmuls: 16 threads was the fastest of
    # 8/16/32/64/128/256 on the 256-core GPU host (tools/cpu_threads.py), so that is what is timed
    threads = min(16, os.cpu_count() or 1)
    try:
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(limits=threads)
    except Exception:                       # pragma: no cover - threadpoolctl is in the image
        limiter = None
        threads = os.cpu_count() or 1
    rng = np.random.RandomState(0)
    p = om.init_params(L, H, D, C, seed=1234, dtype=np.float32)
    m = {k: np.zeros_like(v) for k, v in p.items()}
    v = {k: np.zeros_like(vv) for k, vv in p.items()}
    x = rng.randn(t_sample, B, D).astype(np.float32)
    lengths = np.full(B, t_sample, np.int32)
    dense = np.zeros((B, U), np.int32)
    for b in range(B):
        n = rng.randint(80, 161)
        dense[b, :n - 1] = rng.randint(1, C - 1, size=n - 1)
        dense[b, n - 1] = C - 1
    om.train_step(dict((k, a.copy()) for k, a in p.items()), dict(m), dict(v), 1,
                  [(x[:8], np.full(B, 8, np.int32), dense)], L, 3e-4, 1.0)      # warm the BLAS pool
    t0 = time.time()
    for i in range(steps):
        om.train_step(p, m, v, i + 1, [(x, lengths, dense)], L, 3e-4, 1.0)
    dt = time.time() - t0
    if limiter is not None:
        limiter.restore_original_limits()
    return {"value": steps * B * t_sample / dt, "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": "%d optimiser steps of the numpy oracle (Linear->LSTM stack->Linear->CTC->BPTT->clip+Adam, "
                      "fp32 OpenBLAS, %d threads = fastest setting on this host), %dx%d, B=%d, first %d frames of each utterance, "
                      "%.1f s wall" % (steps, threads, L, H, B, t_sample, dt)}


def cpu_baseline_torch(t_sample=None):
    """the torch-CPU restatement of the same graph (oracle/torch_graph.py: torch's own
    LSTM / CTC CPU kernels, the closest available analogue of TensorFlow-CPU's Eigen/MKL kernels): whole
    optimiser steps on `t_sample` frames per utterance, at the better of two thread counts, ~10 s of work."""
    import torch
    from oracle import model as om
    from oracle.torch_graph import TorchGraph
    t_sample = T if t_sample is None else t_sample
    rng = np.random.RandomState(0)
    p = om.init_params(L, H, D, C, seed=1234, dtype=np.float32)
    x = rng.randn(t_sample, B, D).astype(np.float32)
    lengths = np.full(B, t_sample, np.int32)
    rows = [list(rng.randint(1, C - 1, size=rng.randint(80, 161) - 1)) + [C - 1] for _ in range(B)]
    before = torch.get_num_threads()
    probe = []
    for threads in sorted(set([min(16, os.cpu_count() or 1), min(64, os.cpu_count() or 1)])):
        torch.set_num_threads(threads)
        tg = TorchGraph(p, L)
        tp = min(t_sample, 200)
        t0 = time.time()
        tg.train_step(x[:tp], np.full(B, tp, np.int32), [r[:40] + [C - 1] for r in rows], 3e-4, 1.0)
        probe.append((time.time() - t0, threads))
    threads = min(probe)[1]
    torch.set_num_threads(threads)
    tg = TorchGraph(p, L)
    steps, t0 = 0, time.time()
    while steps < 8 and (steps == 0 or time.time() - t0 < 10.0):
        tg.train_step(x, lengths, rows, 3e-4, 1.0)
        steps += 1
    dt = time.time() - t0
    torch.set_num_threads(before)
    return {"value": steps * B * t_sample / dt, "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": "%d optimiser steps of the torch-CPU restatement (torch LSTM + CTC CPU kernels, fp32, %d threads = "
                      "the faster of 16/64), B=%d, %d frames per utterance, %.1f s wall" % (steps, threads, B, t_sample, dt)}


def dropin_run_train_step(steps, train_decoder="greedy"):
    """extras.dropin_run_train_step: the reference's API path end to end -- AcousticModel.run_train_step over an
    in-memory dataset of raw signals: host batching, PCM upload (20 MB per mini-batch at cfg2), front end, training
    step, loss read-back, greedy decode + merge_repeated + edit distance (the reference decodes on every training
    mini-batch, models/AcousticModel.py:641), the dataflow kernels' status check."""
    import torch
    from models.AcousticModel import AcousticModel, Session
    from models.SpeechRecognizer import SpeechRecognizer
    cm = SpeechRecognizer().get_char_map()
    rng = np.random.RandomState(5)
    words = ["hello", "there", "general", "speech", "recognition", "works", "on", "the", "new", "chip"]
    n = SR * SECONDS
    items = [[(synth_pcm(1000 + i, n), SR), " ".join(rng.choice(words, size=18)), None]
             for i in range(B * (steps + 16))]          # (warm-up steps + timed steps; never an empty step in the timed loop)
    model = AcousticModel(L, H, B, T, U, D, False, len(cm))
    sess = Session()
    ds = model.build_dataset(items, B, T, U, MODE, cm, n_mfcc=D)
    t_it, v_it = model.add_datasets_input(ds, model.build_dataset(items[:B], B, T, U, MODE, cm, n_mfcc=D))
    sess.run(t_it.initializer)
    sess.run(v_it.initializer)
    model.create_training_rnn(0.8, 0.5, 1, 3e-4, 0.33, use_iterator=True)
    model.train_decoder = train_decoder
    if os.environ.get("AMDSPEECH_TRAIN_DECODER_LAG"):      # dev: sweep of the pipeline depth
        model.train_decoder_lag = int(os.environ["AMDSPEECH_TRAIN_DECODER_LAG"])
    for _ in range(5 + model.train_decoder_lag):            # (steady state: the pinned staging pool of the input pipeline fills during the first steps)
        model.run_train_step(sess, 1, 1.0)
    torch.cuda.synchronize()
    import resource
    ru0 = resource.getrusage(resource.RUSAGE_SELF)
    t0 = time.perf_counter()
    for _ in range(steps):
        loss, err, _, _ = model.run_train_step(sess, 1, 1.0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    ru1 = resource.getrusage(resource.RUSAGE_SELF)
    host_cores = ((ru1.ru_utime + ru1.ru_stime) - (ru0.ru_utime + ru0.ru_stime)) / (dt * steps)
    if model._async_beam is not None:
        model._async_beam.close()
    how = ("greedy decode / merge_repeated / edit distance on the GPU" if train_decoder == "greedy" else
           "the reference's width-100 beam decoder + edit distance on host threads, asynchronously (logits by DMA beside the CTC "
           "stage, results %d mini-batches late)" % model.train_decoder_lag)
    return {"value": B * T / dt, "unit": "frames/s", "ms_per_step": dt * 1e3, "steps": steps, "train_decoder": train_decoder,
            "host_cores_busy": round(host_cores, 2),
            "what": "AcousticModel.run_train_step(mini_batch_size=1): host batching + H2D of the PCM + front end + step + "
                    "loss read-back + " + how + " + status check",
            "last_loss": loss, "last_error_rate": err}


def kernel_src_sha16():
    """sha256 (first 16 hex digits) over the sources of the recurrence kernels -- what profiles/rNN_pmc_*.json's figures depend on;
    tools/collect_profiles.sh stamps the same hash into the file it writes."""
    import hashlib
    h = hashlib.sha256()
    for f in ("lstm.hip", "lstm_flow.h", "lstm_flow_fwd.h", "lstm_flow_bwd.h", "lstm_big_fwd.h", "lstm_big_bwd.h", "lstm_step_fwd.h",
              "lstm_step_bwd.h", "lstm_step_bf3.h", "ctc_flow.h", "ctc_core.h", "gemm_core.h", "common.h"):
        with open(os.path.join(ROOT, "rnn-speech_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def main():
    global L, H, D, B, T, MODE
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS),
                    help="cfg2 = BASELINE configs[1] (the headline, default); cfg3 = configs[2] (5x1024, fbank, batch 64)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-frontend", action="store_true", help="time the model step on resident features only")
    ap.add_argument("--precision", default="f32", choices=["f32", "bf16x3", "bf16"],
                    help="arithmetic of the recurrent products for the HEADLINE number (default: exact f32)")
    ap.add_argument("--sync-each-step", action="store_true",
                    help="counter (rocprofv3 --pmc) passes only: bound the number of outstanding dispatches; the "
                         "profiler's queue interceptor faults once several thousand are in flight. Never for timing.")
    ap.add_argument("--no-alt", action="store_true", help="skip the extra measurements (model-only, ragged lengths, drop-in API)")
    ap.add_argument("--ragged", action="store_true",
                    help="also time RAGGED mini-batches (lengths ~U[600, T]) dealt by dataparallel.shard_bucketed (global length "
                         "buckets) and by rank-local buckets: multi_gpu.ragged.  On by default in a multi-rank job")
    ap.add_argument("--no-ragged", action="store_true", help="multi-rank job: skip multi_gpu.ragged")
    ap.add_argument("--alt-bf16x3", action="store_true",
                    help="also time the opt-in split-precision mode (on by default at cfg2 on one GPU unless --no-alt)")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    L, H, D, B, T, MODE = cfg["L"], cfg["H"], cfg["D"], cfg["B"], cfg["T"], cfg["mode"]

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and args.gpus != 1:
        if "WORLD_SIZE" in os.environ:
            sys.exit("bench.py --gpus %d inside a job of %d ranks (torch.distributed.run --nproc-per-node must match)"
                     % (args.gpus, world))
        # Invoked plainly (`python bench.py --gpus N`, the way the driver runs --gpus 1): become the launcher -- one process per
        # GPU under torch.distributed.run on this node, same arguments; rank 0 prints the single JSON line, our exit status is
        # the job's.  (The step loop being timed on every rank: /root/reference/models/AcousticModel.py:887-939.)
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    # AMDSPEECH_BENCH_SHARE_GPU=1 + AMDSPEECH_DIST_BACKEND=gloo: dev-only rehearsal of the multi-rank
    # code path on a 1-GPU box (all ranks on cuda:0, all-reduce staged through the host)
    share_gpu = os.environ.get("AMDSPEECH_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        # the dataflow kernels need the whole GPU for themselves (one resident workgroup per CU for a whole
        # sequence): ranks time-slicing one GPU would run into their bounded waits
        os.environ.setdefault("AMDSPEECH_FLOW", "0")
        os.environ["AMDSPEECH_SHARE_GPU"] = "1"
    torch.cuda.set_device(0 if share_gpu else local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("AMDSPEECH_DIST_BACKEND", "nccl")
        dist.init_process_group(backend)   # bootstrap; nccl == RCCL over xGMI
        if backend == "nccl":
            # the gradient all-reduce MUST be the C-ABI RCCL communicator: a fallback to torch.distributed is an error here,
            # not a log line (the rehearsal on one shared GPU, backend gloo, is the only run that goes through torch)
            os.environ.setdefault("AMDSPEECH_COMM", "rccl")

    import ctypes
    from rnn_speech_amd import lib as _lib
    if not os.path.exists(_lib.LIB_PATH):          # normally prebuilt in-tree by __graft_entry__.build()
        if local == 0:
            import __graft_entry__
            __graft_entry__.build()
        else:
            for _ in range(600):
                if os.path.exists(_lib.LIB_PATH):
                    break
                time.sleep(1.0)
            time.sleep(2.0)
    from rnn_speech_amd import dataparallel, ops
    from rnn_speech_amd.engine import Engine

    grp = dataparallel.current()                   # world > 1: RCCL communicator behind the C ABI + gloo host channel
    BIDIR = bool(cfg.get("bidirectional", False))
    eng = Engine(L, H, D, C, B, T, U, seed=1234, precision=args.precision, bidirectional=BIDIR)   # same seed on every rank
    # the whole job runs on a real (non-NULL) stream (see Engine.on_stream)
    torch.cuda.set_stream(eng.stream)
    n = SR * SECONDS
    n_samples = [n] * B
    # N_ROTATE distinct mini-batches (PCM and labels) resident in HBM, visited round-robin by the timed loop
    pcm_dev, dlab = [], []
    for k in range(N_ROTATE):
        pcm = np.stack([synth_pcm((rank * N_ROTATE + k) * B + b, n) for b in range(B)])
        pcm_dev.append(torch.from_numpy(pcm).cuda())
        dlab.append(torch.from_numpy(synth_labels(np.random.RandomState(100 + rank * N_ROTATE + k), B)).cuda())
    feat, nframes = ops.frontend(pcm_dev[0], n_samples, SR, MODE, T, D)
    assert nframes[0] == T, nframes
    lengths = torch.tensor([min(f, T) for f in nframes], dtype=torch.int32).cuda()

    # Input pipelining, as the reference's tf.data prefetch does: the front end of step k+1 runs on a side stream BESIDE THE
    # FORWARD RECURRENCE of step k, on the two XCDs that kernel leaves idle (Engine.mini_batch(beside_forward=...) ->
    # amdspeech_lstm_beside_forward; rounds 1-2 and AMDSPEECH_BESIDE_FORWARD=0: beside the CTC stage), done long before the
    # backward kernel starts.  One front-end pass per step, in the timed region.
    side = torch.cuda.Stream()
    ahead = {}

    def prefetch_features(i, after=None):
        if after is not None:
            side.wait_event(after)
        with torch.cuda.stream(side):
            f = ops.frontend(pcm_dev[i % N_ROTATE], n_samples, SR, MODE, T, D)[0]
            ev = torch.cuda.Event()
            ev.record(side)
        ahead["f"], ahead["ev"] = f, ev
        return ev

    ar_events = None               # (timed region only: HIP-event pairs around the gradient all-reduce of every step)

    def step(i, e=None):
        e = eng if e is None else e
        if args.no_frontend:
            x, hook = feat, None
        else:
            if "f" not in ahead:
                prefetch_features(i)
            x, ev = ahead.pop("f"), ahead.pop("ev")
            cur = torch.cuda.current_stream()
            cur.wait_event(ev)
            x.record_stream(cur)
            hook = lambda after: prefetch_features(i + 1, after)
        e.zero_grads()
        e.mini_batch(x, lengths, dlab[i % N_ROTATE], 0.8, 0.5, seed=i + 1, beside_forward=hook)
        if world > 1 and ar_events is not None:      # (events on the stream the collective is enqueued on)
            a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a0.record()
            e.all_reduce_grads()
            a1.record()
            ar_events.append((a0, a1))
        else:
            e.all_reduce_grads()
        e.apply(3e-4, 1.0)
        if args.sync_each_step:
            torch.cuda.synchronize()

    def fence():
        torch.cuda.synchronize()
        grp.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    lib = _lib.load()
    _lib.check(lib.amdspeech_profile_enable(1))
    fence()
    ar_events = []
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    fence()
    elapsed = time.perf_counter() - t0
    ar_pairs, ar_events = ar_events, None
    # HIP-event time of the last step's two recurrence launches (recorded on the launch stream)
    ms, nl = ctypes.c_float(), ctypes.c_int()
    _lib.check(lib.amdspeech_profile_get(0, ctypes.byref(ms), ctypes.byref(nl)))
    fwd_ms, time_steps = ms.value, nl.value
    _lib.check(lib.amdspeech_profile_get(1, ctypes.byref(ms), ctypes.byref(nl)))
    bwd_ms = ms.value
    f_rec, f_other = ctypes.c_double(), ctypes.c_double()
    _lib.check(lib.amdspeech_profile_get_flops(1, ctypes.byref(f_rec), ctypes.byref(f_other)))
    bwd_launch_flops = (f_rec.value, f_other.value)      # (0, 0) unless the whole-sequence dataflow kernel ran
    multi = None
    if world > 1:
        # what a disappointing N-GPU number would have to be explained with: which library carried the gradients and how
        # many ranks IT saw, the collective's own time, and the spread of the ranks' clocks
        ar_ms = [a.elapsed_time(b) for a, b in ar_pairs]
        mine = {"rank": rank, "ms_per_step": elapsed / args.steps * 1e3,
                "allreduce_ms": float(np.mean(ar_ms)) if ar_ms else None,
                "allreduce_ms_max": float(np.max(ar_ms)) if ar_ms else None,
                "device": torch.cuda.get_device_name(), "comm": grp.comm_info()}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine, group=grp.host_group)
        info = per_rank[0]["comm"]
        multi = {"device_channel": grp.device_channel,
                 "rccl_ranks": info["world"] if info else None,
                 "rccl_version": info["rccl_version"] if info else None,
                 "rccl_lib": info["lib_path"] if info else None,
                 "rccl_ranks_consistent": bool(info) and all(r["comm"] and r["comm"]["world"] == world and r["comm"]["rank"] == r["rank"]
                                                             for r in per_rank),
                 "allreduce_bytes": int(eng.grads.numel()) * 4,
                 "allreduce_ms": float(np.mean([r["allreduce_ms"] for r in per_rank])),
                 "allreduce_ms_max_over_ranks": float(np.max([r["allreduce_ms_max"] for r in per_rank])),
                 "ms_per_step_min_rank": float(np.min([r["ms_per_step"] for r in per_rank])),
                 "ms_per_step_max_rank": float(np.max([r["ms_per_step"] for r in per_rank])),
                 "what": "allreduce_ms = HIP events around amdspeech_allreduce_sum_f32 on the training stream (includes waiting "
                         "for the slowest rank's backward pass); ms_per_step_*_rank = each rank's own wall clock over the timed loop"}
        tt = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX, group=grp.host_group)
        elapsed = float(tt[0])
    loss = float(eng.loss.mean().cpu())
    eng.check()          # the dataflow kernels' bounded waits: a time-out would have left an error flag

    # Real data is ragged.  The headline above times equal-length utterances (BASELINE's 10 s clips); this section times the same
    # step on mini-batches whose lengths are ~U[600, T], dealt to the ranks in the two ways a data-parallel job can deal them:
    #   global buckets     dataparallel.shard_bucketed -- the WHOLE job's utterances sorted, every bucket of B x world dealt like a
    #                      hand of cards: at every step the ranks' longest utterances are neighbours in the sorted order;
    #   rank-local buckets every rank gets a random equal shard and buckets ITS utterances: every all-reduce waits for the longest
    #                      of `world` unrelated buckets.
    # Reported: valid frames per second of the whole job (max over ranks of the wall clock), the mean spread of the ranks' longest
    # utterance per step, and the drop-in step with the reference's beam decoder on host threads (host cores busy per rank).
    # (size ordering in the reference: /root/reference/models/SpeechRecognizer.py:58-99; global padding: AcousticModel.py:825-827)
    ragged_dp = None
    if (args.ragged or (world > 1 and not args.no_ragged)) and not args.no_frontend:
        pool_n = N_ROTATE * B * world
        pool = np.random.RandomState(77).randint(600, T + 1, size=pool_n)
        items = [[j, None, float(pool[j])] for j in range(pool_n)]
        mine_global = dataparallel.shard_bucketed(items, B, rank, world, seed=1234)
        shard_local = [items[j] for j in np.random.RandomState(78).permutation(pool_n)[rank::world]]
        shard_local.sort(key=lambda it: it[2])
        local_groups = [shard_local[j:j + B] for j in range(0, len(shard_local), B)]
        np.random.RandomState(79 + rank).shuffle(local_groups)
        mine_local = [it for g in local_groups for it in g]

        def run_scheme(mine):
            batches = [np.array([int(it[2]) for it in mine[j:j + B]], np.int32) for j in range(0, len(mine), B)]
            assert len(batches) == N_ROTATE and all(len(v) == B for v in batches)
            dev = [torch.from_numpy(v).cuda() for v in batches]
            longest = [int(v.max()) for v in batches]

            def one(i):
                j = i % N_ROTATE
                x = ops.frontend(pcm_dev[j], n_samples, SR, MODE, T, D)[0]
                eng.zero_grads()
                eng.mini_batch(x, dev[j], dlab[j], 0.8, 0.5, seed=i + 1, max_len=longest[j])
                eng.all_reduce_grads()
                eng.apply(3e-4, 1.0)

            for i in range(args.warmup):
                one(i)
            fence()
            t_r = time.perf_counter()
            for i in range(args.steps):
                one(args.warmup + i)
            fence()
            dt = time.perf_counter() - t_r
            used = [(args.warmup + i) % N_ROTATE for i in range(args.steps)]
            # job-wide: the slowest rank's clock, the frames of all ranks, the spread of the ranks' longest utterance per step
            vec = np.zeros(2 + world * N_ROTATE)
            vec[0] = float(sum(int(batches[j].sum()) for j in used))
            for j in range(N_ROTATE):
                vec[2 + rank * N_ROTATE + j] = longest[j]
            tot = np.asarray(grp.sum_scalars(list(vec)))
            tt_r = torch.tensor([dt], dtype=torch.float64)
            if world > 1:
                dist.all_reduce(tt_r, op=dist.ReduceOp.MAX, group=grp.host_group)
            per_rank = tot[2:].reshape(world, N_ROTATE)
            spread = [float(per_rank[:, j].max() - per_rank[:, j].min()) for j in used]
            wait_frames = [float(per_rank[:, j].max() * world - per_rank[:, j].sum()) for j in used]
            return {"valid_frames_per_s": float(tot[0]) / float(tt_r[0]), "ms_per_step": float(tt_r[0]) / args.steps * 1e3,
                    "valid_frames_per_step": float(tot[0]) / args.steps,
                    "longest_per_rank_spread": float(np.mean(spread)), "longest_per_rank_spread_max": float(np.max(spread)),
                    "frames_run_for_the_slowest_rank_per_step": float(np.mean(wait_frames))}

        ragged_dp = {"lengths": "%d utterances ~U[600, %d] frames, %d mini-batches of %d per rank" % (pool_n, T, N_ROTATE, B),
                     "global_buckets": run_scheme(mine_global), "rank_local_buckets": run_scheme(mine_local),
                     "what": "the timed step (front end, forward, CTC, BPTT, all-reduce, clip + Adam) on ragged mini-batches; every rank "
                             "stops its recurrence at ITS longest utterance, the all-reduce waits for the slowest rank; "
                             "longest_per_rank_spread = mean over the steps of (max - min over the ranks of the longest utterance)"}
        eng.check()
        # the reference's training-time beam decoder on host threads beside the step: how many host cores every rank keeps busy
        # (eight ranks share one host: 8 x this figure has to fit the node's cores)
        try:
            torch.cuda.set_stream(torch.cuda.default_stream())
            mine_beam = dropin_run_train_step(max(4, min(args.steps, 8)), train_decoder="beam")
            torch.cuda.set_stream(eng.stream)
            cores = grp.sum_scalars([mine_beam["host_cores_busy"] if r == rank else 0.0 for r in range(world)])
            steps_ms = grp.sum_scalars([mine_beam["ms_per_step"] if r == rank else 0.0 for r in range(world)])
            ragged_dp["dropin_beam_decoder"] = {"host_cores_busy_per_rank": [round(float(v), 2) for v in cores],
                                                "ms_per_step_per_rank": [round(float(v), 3) for v in steps_ms],
                                                "what": "AcousticModel.run_train_step (equal-length clips) with train_decoder = beam on every rank"}
        except Exception as exc:       # (host-side extra: every rank fails or none -- same code, same data)
            ragged_dp["dropin_beam_decoder"] = {"error": repr(exc)[:300]}

    # SURVEY 8(d) extras, reported beside the headline (never instead of it): the model step with the
    # front end excluded, a batch with ragged lengths ~U[600,T] (masking + early stop at the longest), and the
    # drop-in API path (AcousticModel.run_train_step from raw host signals)
    extras = None
    if world == 1 and not args.no_alt and not args.no_frontend:
        def timed(fn):
            for i in range(args.warmup):
                fn(i)
            fence()
            t = time.perf_counter()
            for i in range(args.steps):
                fn(args.warmup + i)
            fence()
            return (time.perf_counter() - t) / args.steps

        def model_only(i):
            eng.zero_grads()
            eng.mini_batch(feat, lengths, dlab[i % N_ROTATE], 0.8, 0.5, seed=i + 1)
            eng.apply(3e-4, 1.0)

        rag = np.random.RandomState(7).randint(600, T + 1, size=B).astype(np.int32)
        rag_dev, rag_max = torch.from_numpy(rag).cuda(), int(rag.max())

        def ragged(i):
            x = ops.frontend(pcm_dev[i % N_ROTATE], n_samples, SR, MODE, T, D)[0]
            eng.zero_grads()
            eng.mini_batch(x, rag_dev, dlab[i % N_ROTATE], 0.8, 0.5, seed=i + 1, max_len=rag_max)
            eng.apply(3e-4, 1.0)

        # ... and the same kind of utterances LENGTH-BUCKETED (what `dataset_size_ordering : Bucketed` does, and under data
        # parallelism dataparallel.shard_bucketed for the whole job): N_ROTATE x B lengths ~U[600, T], sorted, cut into N_ROTATE
        # mini-batches; a step runs to ITS bucket's longest utterance, so consecutive steps have different T (the workspace is one
        # allocation laid out per length: ops.LstmWorkspace.prefix)
        pool = np.sort(np.concatenate([np.random.RandomState(7 + j).randint(600, T + 1, size=B) for j in range(N_ROTATE)]))
        bk = [pool[j * B:(j + 1) * B].astype(np.int32) for j in range(N_ROTATE)]
        bk_dev, bk_max = [torch.from_numpy(v).cuda() for v in bk], [int(v.max()) for v in bk]

        def bucketed(i):
            j = (i * 3) % N_ROTATE                   # (not in order of length: T goes up and down from step to step)
            x = ops.frontend(pcm_dev[j], n_samples, SR, MODE, T, D)[0]
            eng.zero_grads()
            eng.mini_batch(x, bk_dev[j], dlab[j], 0.8, 0.5, seed=i + 1, max_len=bk_max[j])
            eng.apply(3e-4, 1.0)

        dt_m, dt_r, dt_b = timed(model_only), timed(ragged), timed(bucketed)
        # the same model-only step with the CTC stage as separate launches between the two recurrence kernels (rounds 1 - 4)
        from rnn_speech_amd import engine as _engine_mod
        fused_now = getattr(eng, "_head", None) is not None
        _old_fused, _engine_mod._FUSED_CTC = _engine_mod._FUSED_CTC, False
        try:
            dt_sep = timed(model_only)
        finally:
            _engine_mod._FUSED_CTC = _old_fused
        model_only(0)                    # (leave the engine on the default path)
        bk_valid = sum(int(bk[((args.warmup + i) * 3) % N_ROTATE].sum()) for i in range(args.steps))
        extras = {"model_only_frontend_excluded": {"value": B * T / dt_m, "unit": "frames/s", "ms_per_step": dt_m * 1e3,
                                                   "ctc_stage": "inside the LSTM launches" if fused_now else "separate launches"},
                  "model_only_separate_ctc_launches": {"value": B * T / dt_sep, "unit": "frames/s", "ms_per_step": dt_sep * 1e3,
                                                       "what": "AMDSPEECH_FUSED_CTC=0: output Linear, log-softmax, alpha / beta, gradient and "
                                                               "dlogits . W_o^T as launches between the two recurrence kernels"},
                  "ragged_lengths_u600_T": {"value": float(rag.sum()) / dt_r, "unit": "valid frames/s",
                                            "ms_per_step": dt_r * 1e3, "valid_frames": int(rag.sum()),
                                            "longest": rag_max},
                  "ragged_lengths_u600_T_bucketed": {"value": bk_valid / (dt_b * args.steps), "unit": "valid frames/s",
                                                     "ms_per_step": dt_b * 1e3, "valid_frames_per_step": bk_valid / args.steps,
                                                     "longest_per_bucket": bk_max,
                                                     "what": "%d x %d lengths ~U[600, T] sorted into %d mini-batches; each step stops "
                                                             "at its bucket's longest utterance" % (N_ROTATE, B, N_ROTATE)}}
        # RCCL and the dataflow kernels in ONE driver-timed process: a C-ABI communicator with a world of one rank,
        # amdspeech_allreduce_sum_f32 over the real flat gradient buffer on the training stream between the backward tail and
        # clip + Adam -- the launch-overhead baseline the first real multi-GPU run is to be compared with (the exchange itself:
        # /root/reference/models/AcousticModel.py:391-406, one accumulation per optimiser step)
        def rccl_world1():
            import ctypes as C_
            ident = (C_.c_char * _lib.COMM_ID_BYTES)()
            comm = C_.c_void_p()
            if lib.amdspeech_comm_unique_id(ident) != 0 or lib.amdspeech_comm_init(ident, 0, 1, C_.byref(comm)) != 0:
                return {"error": lib.amdspeech_last_error().decode("utf-8", "replace")[:300]}
            try:
                r, w, v = C_.c_int(), C_.c_int(), C_.c_int()
                path = C_.create_string_buffer(512)
                _lib.check(lib.amdspeech_comm_info(comm, C_.byref(r), C_.byref(w), C_.byref(v), path, 512), "comm_info")
                pairs = []

                def one(i):
                    eng.zero_grads()
                    eng.mini_batch(feat, lengths, dlab[i % N_ROTATE], 0.8, 0.5, seed=i + 1)
                    a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a0.record()
                    _lib.check(lib.amdspeech_allreduce_sum_f32(comm, C_.c_void_p(torch.cuda.current_stream().cuda_stream),
                                                               C_.c_void_p(eng.grads.data_ptr()), eng.grads.numel()), "allreduce_sum_f32")
                    a1.record()
                    pairs.append((a0, a1))
                    eng.apply(3e-4, 1.0)

                dt = timed(one)
                ar = [a.elapsed_time(b) for a, b in pairs[args.warmup:]]
                eng.check()
                return {"ms_per_step": dt * 1e3, "ms_per_step_without": dt_m * 1e3, "allreduce_ms": float(np.mean(ar)),
                        "allreduce_ms_max": float(np.max(ar)), "allreduce_bytes": int(eng.grads.numel()) * 4,
                        "rccl_version": v.value, "rccl_lib": path.value.decode(), "rccl_ranks": w.value,
                        "dataflow_kernels": os.environ.get("AMDSPEECH_FLOW", "1") != "0",
                        "what": "the model-only step (front end excluded) with amdspeech_allreduce_sum_f32 of the flat gradient "
                                "buffer, world of ONE rank, on the training stream between the backward tail and clip + Adam; "
                                "allreduce_ms = HIP events around the call"}
            finally:
                lib.amdspeech_comm_destroy(comm)

        try:
            extras["rccl_world1"] = rccl_world1()
        except Exception as exc:       # the headline must not die with the extra
            extras["rccl_world1"] = {"error": repr(exc)[:300]}
        torch.cuda.set_stream(torch.cuda.default_stream())
        extras["dropin_run_train_step"] = dropin_run_train_step(max(4, min(args.steps, 10)))
        extras["dropin_run_train_step_beam"] = dropin_run_train_step(max(8, min(args.steps, 20)), train_decoder="beam")
        torch.cuda.set_stream(eng.stream)

    # BASELINE configs[2] (5x1024, 120-dim fbank + deltas, batch 64) in the same default run, so that it is driver-measured:
    # a child process runs this script with --config cfg3 for 3 steps (this process idles meanwhile) and its line is embedded
    if extras is not None and args.config == "cfg2" and os.environ.get("AMDSPEECH_BENCH_CFG3", "1") != "0":
        import subprocess
        torch.cuda.synchronize()
        try:
            child = subprocess.run([sys.executable, os.path.abspath(__file__), "--config", "cfg3", "--steps", "3", "--warmup", "1",
                                    "--no-alt", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600)
            line = [ln for ln in child.stdout.splitlines() if ln.startswith("{")][-1]
            c3 = json.loads(line)
            extras["cfg3"] = {"metric": c3["metric"], "value": c3["value"], "unit": c3["unit"], "ms_per_step": c3["ms_per_step"],
                              "steps": c3["steps"], "warmup": c3["warmup"], "workload": c3["config"]["workload"],
                              "fp32_mfma_ceiling_ms": 102.7,
                              "roofline": {k: c3["roofline"][k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac",
                                                                          "avg_time_step_us", "flops_per_time_step", "launch_ms")}}
        except Exception as exc:       # the headline must not die with the extra
            extras["cfg3"] = {"error": repr(exc)[:300]}
        # the opt-in split-precision mode at the two H = 1024 configurations (never the headline: alt_*): recurrent AND batched
        # products as bf16 hi/lo pairs on the bf16 MFMA (BASELINE configs[4] asks for "bf16 MFMA")
        # ... and (round 4) as PLAIN bf16 operands, one MFMA per product (alt_bf16_*): f32 accumulation, master weights, gates, state
        for tag, cfg_name, prec in (("alt_bf16x3_cfg3", "cfg3", "bf16x3"), ("alt_bf16x3_cfg5_bidirectional", "cfg5", "bf16x3"),
                                    ("alt_bf16_cfg3", "cfg3", "bf16"), ("alt_bf16_cfg5_bidirectional", "cfg5", "bf16")):
            try:
                child = subprocess.run([sys.executable, os.path.abspath(__file__), "--config", cfg_name, "--steps", "3", "--warmup", "1",
                                        "--no-alt", "--no-cpu-baseline", "--precision", prec], capture_output=True, text=True, timeout=600)
                c3 = json.loads([ln for ln in child.stdout.splitlines() if ln.startswith("{")][-1])
                extras[tag] = {"metric": c3["metric"], "value": c3["value"], "unit": c3["unit"], "ms_per_step": c3["ms_per_step"],
                               "steps": c3["steps"], "dtype": c3["dtype"], "workload": c3["config"]["workload"]}
            except Exception as exc:
                extras[tag] = {"error": repr(exc)[:300]}

    # separately reported: the opt-in split-precision mode (NOT the headline)
    alt = None
    if args.precision == "f32" and (args.alt_bf16x3 or (world == 1 and not args.no_alt and args.config == "cfg2")):
        eng3 = Engine(L, H, D, C, B, T, U, seed=1234, precision="bf16x3", bidirectional=BIDIR)
        ref_logits = Engine(L, H, D, C, B, T, U, seed=1234, bidirectional=BIDIR).forward(feat, lengths).clone()
        diff = float(((eng3.forward(feat, lengths) - ref_logits).abs().max() / ref_logits.abs().max()).cpu())
        for i in range(args.warmup):
            step(i, eng3)
        fence()
        t1 = time.perf_counter()
        for i in range(args.steps):
            step(args.warmup + i, eng3)
        fence()
        el3 = time.perf_counter() - t1
        alt = {"precision": "bf16x3 (the recurrent products as hi.hi + hi.lo + lo.hi on v_mfma_f32_16x16x32_bf16 inside the same "
                            "dataflow kernels, f32 accumulate, everything else f32; opt-in, not the headline)",
               "value": B * T * world / (el3 / args.steps), "unit": "frames/s", "ms_per_step": el3 / args.steps * 1e3,
               "logits_max_rel_diff_vs_f32_path": diff}

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        frames = B * T * world
        # dominant kernel = the BPTT recurrence (largest share of the step in profiles/): algorithmic FLOPs per time
        # step = (2L-1) products [B,4H]x[4H,H] (GEMM flops, 2/MAC); one launch of a dataflow kernel covers
        # `time_steps` = T + L - 1 of them, the launch-per-diagonal kernels one
        bwd_flops = (2 * L - 1) * 2.0 * B * 4 * H * H
        fwd_flops = L * 2.0 * B * 2 * H * 4 * H
        layerwise = time_steps == T * L      # H = 1024: one launch per layer, x / down products hoisted into GEMMs
        if layerwise:                        # a time step of ONE layer: only the recurrent product is left in the kernel
            bwd_flops = fwd_flops = 2.0 * B * 4 * H * H
        bwd_us = bwd_ms * 1e3 / time_steps
        rec_only = None
        if bwd_launch_flops[0] > 0 and not layerwise:
            # the dataflow launch as the library accounts for it: the recurrence's products (incl. dZ_0 when the bottom layer's
            # groups form it) PLUS the weight-gradient products its worker workgroups compute in the same launch
            rec_only = {"flops_per_time_step": bwd_launch_flops[0] / time_steps,
                        "achieved": bwd_launch_flops[0] / (bwd_ms * 1e-3) / 1e12,
                        "frac": bwd_launch_flops[0] / (bwd_ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                        "what": "the recurrence's own products only (the in-kernel weight-gradient share left out)"}
            bwd_flops = (bwd_launch_flops[0] + bwd_launch_flops[1]) / time_steps
        achieved = bwd_flops / (bwd_us * 1e-6) / 1e12
        # from the committed rocprofv3 PMC passes of THIS command (separate --pmc runs, tools/collect_profiles.sh):
        # HBM-side bytes per time step (FETCH_SIZE doubled per MI355X_MICROARCH.md 'HBM') and the measured MFMA-pipe
        # utilisation of the kernel (SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CU_CYCLES)
        traffic = mfma_util = None
        tag = None
        profile_src = None           # hash of the kernel sources the committed counters were collected on (tools/collect_profiles.sh)
        for tag_try in ("r06", "r05", "r04", "r03", "r02", "r01"):
            pmc = os.path.join(ROOT, "profiles", "%s_pmc_%s.json" % (tag_try, args.config))
            if not os.path.exists(pmc) and args.config == "cfg2":
                pmc = os.path.join(ROOT, "profiles", "%s_pmc_fetch_write_size.json" % tag_try)
            if os.path.exists(pmc):
                tag = os.path.basename(pmc)
                blob = json.load(open(pmc))
                profile_src = blob.get("_meta", {}).get("kernel_src_sha16")
                for name, c in blob.items():
                    if "lstm_bwd" not in name:
                        continue
                    per = time_steps if "flow" in name else (T if "big" in name else 1)      # (a per-layer launch covers T steps)
                    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                        traffic = (2.0 * c["FETCH_SIZE"]["median"] + c["WRITE_SIZE"]["median"]) * 1024.0 / per
                    if "mfma_util" in c:
                        mfma_util = c["mfma_util"]
                break
        out = {
            "metric": cfg["metric"],
            "value": frames / (elapsed / args.steps),
            "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s from 10 s / 16 kHz synthetic PCM resident in HBM (%d distinct mini-batches in "
                                   "rotation), batch %d per GPU, T=%d frames, dropout keep 0.8/0.5, clip 1 + Adam; front end %s"
                                   % (cfg["name"], N_ROTATE, B, T,
                                      "excluded from the timed step" if args.no_frontend else
                                      "inside the timed step (one pass per step; the pass for step k+1 runs on a side stream beside "
                                      "step k's forward recurrence where that kernel leaves XCDs idle, else beside its CTC stage); "
                                      "CTC stage %s" % ("inside the two LSTM launches (ctc_flow.h)" if getattr(eng, "_head", None) is not None
                                                        else "as separate launches")),
                       "global_batch": B * world, "frames_per_step": frames, "parallelism": "dp%d" % world,
                       "mean_ctc_loss": loss, "fwd_recurrence_ms": fwd_ms, "bwd_recurrence_ms": bwd_ms,
                       "time_steps": time_steps,
                       "device_channel": grp.device_channel,
                       "rccl_ranks": multi["rccl_ranks"] if multi else None},
            "roofline": {"kernel": ("BPTT recurrence, one layer per launch (lstm_bwd_big; figures per time step of one layer: the "
                                    "recurrent product only, the other products are hoisted into GEMMs)" if layerwise else
                                    "BPTT recurrence launch (lstm_bwd_flow2; figures per time step of the whole stack: recurrent + down "
                                    "products of every layer, dZ_0, and the weight-gradient products of the in-kernel GEMM workers)"),
                         "bound": "mfma",
                         "achieved": achieved, "peak": FP32_MFMA_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / FP32_MFMA_PEAK_TFLOPS, "traffic": traffic,
                         "traffic_unit": "bytes per time step (2*FETCH_SIZE + WRITE_SIZE of the launch / time steps, profiles/%s)" % tag,
                         # the counters are a committed collection, not this run's: both source hashes, so a stale figure shows
                         "traffic_profile": {"file": tag, "kernel_src_sha16": profile_src, "this_build_kernel_src_sha16": kernel_src_sha16(),
                                             "stale": (profile_src != kernel_src_sha16()) if profile_src is not None else None},
                         "mfma_util_measured": mfma_util,
                         "avg_time_step_us": bwd_us, "flops_per_time_step": bwd_flops, "launch_ms": bwd_ms, "recurrence_only": rec_only,
                         "fwd_step": {"avg_time_step_us": fwd_ms * 1e3 / time_steps,
                                      "achieved": fwd_flops / (fwd_ms * 1e-3 / time_steps) / 1e12}},
        }
        if multi is not None:
            out["multi_gpu"] = multi
        if ragged_dp is not None:
            if multi is not None:
                out["multi_gpu"]["ragged"] = ragged_dp
            else:
                out["ragged"] = ragged_dp
        if extras is not None:
            out["extras"] = extras
        if alt is not None:
            out["alt_bf16x3"] = alt
        if args.precision != "f32":
            out["dtype"] = "f32 storage, %s MFMA products (opt-in mode)" % args.precision
        if not args.no_cpu_baseline and world == 1 and not BIDIR:      # (the CPU restatements timed here are unidirectional)
            # bounded sample: ~10-30 s of CPU work whatever the configuration
            t_s = T if args.config == "cfg2" else 120
            a, b = cpu_baseline(t_s), cpu_baseline_torch(t_s)
            # the faster restatement is THE baseline; the other is kept beside it
            out["cpu_baseline"], out["cpu_baseline_other"] = (a, b) if a["value"] >= b["value"] else (b, a)
        print(json.dumps(out))
    if world > 1:
        grp.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
