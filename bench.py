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

L, H, D, C, B, T, U = 3, 512, 40, 80, 32, 1001, 161
SR, SECONDS = 16000, 10
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


def cpu_baseline(t_sample=T, steps=2):
    """The numpy oracle (a PORT of the reference graph; TensorFlow is not installable here)
    timed on this host: one optimiser step, B=32, on the first `t_sample` frames."""
    from oracle import model as om
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
            "sample": "%d optimiser steps of the numpy oracle (Linear->3x512 LSTM->Linear->CTC->BPTT->clip+Adam, "
                      "fp32 OpenBLAS, %d threads = fastest setting on this host), B=%d, first %d frames of each utterance, "
                      "%.1f s wall" % (steps, threads, B, t_sample, dt)}


def cpu_baseline_torch(t_sample=T):
    """the torch-CPU restatement of the same graph (oracle/torch_graph.py: torch's own
    LSTM / CTC CPU kernels, the closest available analogue of TensorFlow-CPU's Eigen/MKL kernels): whole
    optimiser steps on `t_sample` frames per utterance, at the better of two thread counts, ~10 s of work."""
    import torch
    from oracle import model as om
    from oracle.torch_graph import TorchGraph
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-frontend", action="store_true", help="time the model step on resident features only")
    ap.add_argument("--precision", default="f32", choices=["f32", "bf16x3"],
                    help="arithmetic of the recurrent products for the HEADLINE number (default: exact f32)")
    ap.add_argument("--sync-each-step", action="store_true",
                    help="counter (rocprofv3 --pmc) passes only: bound the number of outstanding dispatches; the "
                         "profiler's queue interceptor faults once several thousand are in flight. Never for timing.")
    ap.add_argument("--no-alt", action="store_true", help="skip the extra measurements (model-only, ragged lengths)")
    ap.add_argument("--alt-bf16x3", action="store_true",
                    help="also time the opt-in split-precision mode (step-kernel path; no longer faster than the default)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if args.gpus != 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d"
                     % (args.gpus, args.gpus))
    # AMDSPEECH_BENCH_SHARE_GPU=1 + AMDSPEECH_DIST_BACKEND=gloo: dev-only rehearsal of the multi-rank
    # code path on a 1-GPU box (all ranks on cuda:0, all-reduce staged through the host)
    share_gpu = os.environ.get("AMDSPEECH_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        # the dataflow kernels need the whole GPU for themselves (one resident workgroup per CU for a whole
        # sequence): ranks time-slicing one GPU would run into their bounded waits
        os.environ.setdefault("AMDSPEECH_FLOW", "0")
    torch.cuda.set_device(0 if share_gpu else local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(os.environ.get("AMDSPEECH_DIST_BACKEND", "nccl"))   # nccl == RCCL over xGMI

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
    from rnn_speech_amd import ops
    from rnn_speech_amd.engine import Engine
    from rnn_speech_amd.audioprocessor import AudioProcessor

    eng = Engine(L, H, D, C, B, T, U, seed=1234, precision=args.precision)   # same seed on every rank: identical replicas
    # the whole job runs on a real (non-NULL) stream: lets lstm_bwd overlap the weight-gradient GEMMs with the
    # BPTT chain on CU-partitioned streams (blocking streams synchronise implicitly with the legacy NULL stream)
    torch.cuda.set_stream(eng.stream)
    audio = AudioProcessor(T, "mfcc", n_mfcc=D)
    n = SR * SECONDS
    pcm = np.stack([synth_pcm(rank * B + b, n) for b in range(B)])
    pcm_dev = torch.from_numpy(pcm).cuda()
    n_samples = [n] * B
    rng = np.random.RandomState(100 + rank)
    dlab = torch.from_numpy(synth_labels(rng, B)).cuda()
    feat, nframes = ops.frontend(pcm_dev, n_samples, SR, "mfcc", T, D)
    assert nframes[0] == T, nframes
    lengths = torch.tensor([min(f, T) for f in nframes], dtype=torch.int32).cuda()

    # Input pipelining, as the reference's tf.data prefetch does: the front end of step k+1 is enqueued on a side
    # stream right after step k's forward/backward have been enqueued, so it runs under the tail of step k (weight-
    # gradient GEMMs, all-reduce, Adam).  One front-end pass per step, inside the timed region.
    side = torch.cuda.Stream()
    ahead = {}

    def prefetch_features():
        with torch.cuda.stream(side):
            f = ops.frontend(pcm_dev, n_samples, SR, "mfcc", T, D)[0]
            ev = torch.cuda.Event()
            ev.record(side)
        ahead["f"], ahead["ev"] = f, ev

    def step(i, e=None):
        e = eng if e is None else e
        if args.no_frontend:
            x = feat
        else:
            if "f" not in ahead:
                prefetch_features()
            x, ev = ahead.pop("f"), ahead.pop("ev")
            cur = torch.cuda.current_stream()
            cur.wait_event(ev)
            x.record_stream(cur)
        e.zero_grads()
        e.mini_batch(x, lengths, dlab, 0.8, 0.5, seed=i + 1)
        if not args.no_frontend:
            prefetch_features()
        e.all_reduce_grads()
        e.apply(3e-4, 1.0)
        if args.sync_each_step:
            torch.cuda.synchronize()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    lib = _lib.load()
    _lib.check(lib.amdspeech_profile_enable(1))
    fence()
    t0 = time.perf_counter()
    fwd_ms = bwd_ms = 0.0
    launches = 0
    for i in range(args.steps):
        step(args.warmup + i)
    fence()
    elapsed = time.perf_counter() - t0
    # HIP-event time of the last step's two recurrent launch chains (recorded on the launch stream)
    ms, nl = ctypes.c_float(), ctypes.c_int()
    _lib.check(lib.amdspeech_profile_get(0, ctypes.byref(ms), ctypes.byref(nl)))
    fwd_ms, launches = ms.value, nl.value
    _lib.check(lib.amdspeech_profile_get(1, ctypes.byref(ms), ctypes.byref(nl)))
    bwd_ms = ms.value
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64).cuda()
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.cpu())
    loss = float(eng.loss.mean().cpu())
    eng.check()          # the dataflow kernels' bounded waits: a time-out would have left an error flag

    # SURVEY 8(d) extras, reported beside the headline (never instead of it): the model step with the
    # front end excluded, and a batch with ragged lengths ~U[600,1001] (masking + early stop at the longest)
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
            eng.mini_batch(feat, lengths, dlab, 0.8, 0.5, seed=i + 1)
            eng.apply(3e-4, 1.0)

        rag = np.random.RandomState(7).randint(600, T + 1, size=B).astype(np.int32)
        rag_dev, rag_max = torch.from_numpy(rag).cuda(), int(rag.max())

        def ragged(i):
            x = ops.frontend(pcm_dev, n_samples, SR, "mfcc", T, D)[0]
            eng.zero_grads()
            eng.mini_batch(x, rag_dev, dlab, 0.8, 0.5, seed=i + 1, max_len=rag_max)
            eng.apply(3e-4, 1.0)

        dt_m, dt_r = timed(model_only), timed(ragged)
        extras = {"model_only_frontend_excluded": {"value": B * T / dt_m, "unit": "frames/s", "ms_per_step": dt_m * 1e3},
                  "ragged_lengths_u600_1001": {"value": float(rag.sum()) / dt_r, "unit": "valid frames/s",
                                               "ms_per_step": dt_r * 1e3, "valid_frames": int(rag.sum()),
                                               "longest": rag_max}}

    # separately reported: the opt-in split-precision mode (NOT the headline)
    alt = None
    if args.precision == "f32" and args.alt_bf16x3:
        eng3 = Engine(L, H, D, C, B, T, U, seed=1234, precision="bf16x3")
        ref_logits = Engine(L, H, D, C, B, T, U, seed=1234).forward(feat, lengths).clone()
        diff = float(((eng3.forward(feat, lengths) - ref_logits).abs().max() / ref_logits.abs().max()).cpu())
        for i in range(args.warmup):
            step(i, eng3)
        fence()
        t1 = time.perf_counter()
        for i in range(args.steps):
            step(args.warmup + i, eng3)
        fence()
        el3 = time.perf_counter() - t1
        if world > 1:
            tt = torch.tensor([el3], dtype=torch.float64).cuda()
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el3 = float(tt.cpu())
        alt = {"precision": "bf16x3 (hi.hi + hi.lo + lo.hi on bf16 MFMA, f32 accumulate; opt-in, not the headline)",
               "value": B * T * world / (el3 / args.steps), "unit": "frames/s", "ms_per_step": el3 / args.steps * 1e3,
               "logits_max_rel_diff_vs_f32_path": diff}

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        frames = B * T * world
        # dominant kernel = the BPTT diagonal step (largest share of the step in profiles/):
        # algorithmic FLOPs per launch = (2L-1) products [B,4H]x[4H,H] (GEMM flops, 2/MAC)
        bwd_flops = (2 * L - 1) * 2.0 * B * 4 * H * H
        bwd_us = bwd_ms * 1e3 / launches
        achieved = bwd_flops / (bwd_us * 1e-6) / 1e12
        # HBM-side bytes per launch of that kernel from the committed rocprofv3 PMC passes of THIS command
        # (separate --pmc FETCH_SIZE / WRITE_SIZE runs; FETCH_SIZE doubled per MI355X_MICROARCH.md 'HBM')
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r01_pmc_fetch_write_size.json")
        if os.path.exists(pmc):
            for name, c in json.load(open(pmc)).items():
                if "lstm_bwd_flow" in name and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                    # one launch runs the whole sequence: per time step like `achieved`
                    traffic = (2.0 * c["FETCH_SIZE"]["median"] + c["WRITE_SIZE"]["median"]) * 1024.0 / launches
                elif "lstm_bwd_step" in name and traffic is None and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                    traffic = (2.0 * c["FETCH_SIZE"]["median"] + c["WRITE_SIZE"]["median"]) * 1024.0
        out = {
            "metric": "audio_frames_per_sec_train_3x512_lstm_ctc",
            "value": frames / (elapsed / args.steps),
            "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[1]: 3x512 LSTM + CTC training step, 40-dim MFCC from 10 s / 16 kHz "
                                   "synthetic PCM resident in HBM, batch 32 per GPU, T=1001 frames, dropout keep 0.8/0.5, "
                                   "clip 1 + Adam; front end %s" % ("excluded from the timed step" if args.no_frontend else
                                                                    "inside the timed step (one pass per step; the pass for step k+1 is "
                                                                    "enqueued on a side stream under the tail of step k)"),
                       "global_batch": B * world, "frames_per_step": frames, "parallelism": "dp%d" % world,
                       "mean_ctc_loss": loss, "fwd_chain_ms": fwd_ms, "bwd_chain_ms": bwd_ms,
                       "step_launches_per_chain": launches},
            "roofline": {"kernel": "lstm_bwd_flow (BPTT recurrence, one launch per sequence; figures per time step)", "bound": "mfma", "achieved": achieved, "peak": FP32_MFMA_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / FP32_MFMA_PEAK_TFLOPS, "traffic": traffic,
                         "traffic_unit": "bytes per time step (2*FETCH_SIZE + WRITE_SIZE of the launch / steps, profiles/r01_pmc_fetch_write_size.json)",
                         "avg_launch_us": bwd_us, "flops_per_launch": bwd_flops,
                         "fwd_step": {"avg_launch_us": fwd_ms * 1e3 / launches,
                                      "achieved": L * 2.0 * B * 2 * H * 4 * H / (fwd_ms * 1e-3 / launches) / 1e12}},
        }
        if extras is not None:
            out["extras"] = extras
        if alt is not None:
            out["alt_bf16x3"] = alt
        if args.precision != "f32":
            out["dtype"] = "f32 storage, bf16x3 MFMA products (opt-in mode)"
        if not args.no_cpu_baseline and world == 1:
            a, b = cpu_baseline(), cpu_baseline_torch()
            # the faster restatement is THE baseline; the other is kept beside it
            out["cpu_baseline"], out["cpu_baseline_other"] = (a, b) if a["value"] >= b["value"] else (b, a)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
