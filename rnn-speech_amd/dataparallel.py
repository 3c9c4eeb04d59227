"""Data-parallel plumbing of the training path: one process per GPU, utterances sharded by rank, ONE fp32 SUM
all-reduce of the flat gradient buffer per optimiser step, identical clip + Adam on every rank (SURVEY.md 8e;
equivalent to the reference's `mini_batch_size = N` gradient accumulation, models/AcousticModel.py:391-406,
916-926).

Two channels:
  * device: the gradient all-reduce and the parameter broadcast go through the C ABI
    (amdspeech_allreduce_sum_f32 / amdspeech_broadcast_f32 = RCCL over xGMI).  torch.distributed is only the
    bootstrap that carries the 128-byte RCCL id from rank 0 to the others;
  * host: a few scalars per step (has-data flags, the three logging sums, the learning-rate decision) travel
    over a gloo group, so they never touch a device stream.

On CPU tensors (the multi-process tests in this repository run on gloo without a GPU) the device channel falls
back to torch.distributed's own all_reduce of the same flat buffer.
"""
import ctypes as C
import os

import torch

from . import lib as _l

_group = None


class Group(object):
    """The communicator of one training job.  `Group.single()` is the world-size-1 case (every call is a no-op)."""

    def __init__(self, rank, world, host_group=None, comm=None):
        self.rank, self.world = rank, world
        self.host_group = host_group        # gloo process group for host scalars (None when world == 1)
        self._comm = comm                   # void* of the RCCL communicator behind the C ABI (None: torch fallback)

    @classmethod
    def single(cls):
        return cls(0, 1)

    # ---- construction ------------------------------------------------------------------------------------
    @classmethod
    def from_env(cls, device_channel="auto"):
        """Join the job described by RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torch.distributed.run).
        device_channel: "rccl" (C ABI), "torch" (torch.distributed on the default backend) or "auto" (rccl when the
        default backend is nccl, i.e. on GPUs)."""
        import torch.distributed as dist
        world = int(os.environ.get("WORLD_SIZE", "1"))
        if world <= 1:
            return cls.single()
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if not dist.is_initialized():
            backend = os.environ.get("AMDSPEECH_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
            if backend == "nccl" and os.environ.get("AMDSPEECH_SHARE_GPU") != "1":
                torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
            dist.init_process_group(backend)
        rank = dist.get_rank()
        host = dist.new_group(backend="gloo") if dist.get_backend() != "gloo" else dist.group.WORLD
        comm = None
        device_channel = os.environ.get("AMDSPEECH_COMM", device_channel)      # "torch": skip the C-ABI communicator
        strict = device_channel == "rccl"       # asked for by name: no silent fallback (bench.py --gpus N does)
        if device_channel == "rccl" or (device_channel == "auto" and dist.get_backend() == "nccl"):
            comm, why = cls._rccl_init(rank, world, host)
            # every rank must end up on the same channel
            ok = torch.tensor([1 if comm is not None else 0], dtype=torch.int32)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=host)
            if int(ok[0]) == 0:
                if comm is not None:
                    _l.load().amdspeech_comm_destroy(comm)
                    comm = None
                msg = "C-ABI RCCL communicator unavailable on rank %d (%s)" % (rank, why or "another rank failed")
                if strict:
                    raise _l.AmdSpeechError(msg + "; AMDSPEECH_COMM=rccl forbids the torch.distributed fallback")
                import logging
                logging.warning("%s: gradients go through torch.distributed's %s backend", msg, dist.get_backend())
        return cls(rank, world, host, comm)

    @staticmethod
    def _rccl_init(rank, world, host):
        """-> (comm or None, reason).  Never raises and never leaves a peer alone in a collective.  ncclCommInitRank is itself a
        collective, so the ranks first AGREE over the host group that every one of them can bind RCCL (rank 0 makes the unique
        id, the others only check that the library and its symbols are there: amdspeech_comm_available) and nobody enters it
        unless all can; then rank 0 ALWAYS broadcasts its id."""
        import torch
        import torch.distributed as dist
        lib = _l.load()
        ident = (C.c_char * _l.COMM_ID_BYTES)()
        why = None
        # only rank 0 makes an id (ncclGetUniqueId opens a bootstrap listener and a thread: on the other ranks they would never
        # be connected to and live as long as the process); everybody else asks whether RCCL can be bound at all
        rc = lib.amdspeech_comm_unique_id(ident) if rank == 0 else lib.amdspeech_comm_available()
        if rc != 0:
            why = lib.amdspeech_last_error().decode("utf-8", "replace") or "RCCL cannot be bound"
        able = torch.tensor([0 if why else 1], dtype=torch.int32)
        dist.all_reduce(able, op=dist.ReduceOp.MIN, group=host)
        if int(able[0]) == 0:
            return None, why or "another rank cannot bind RCCL"
        box = [bytes(ident.raw) if rank == 0 else None]
        dist.broadcast_object_list(box, src=0, group=host)
        ident = (C.c_char * _l.COMM_ID_BYTES).from_buffer_copy(box[0])
        comm = C.c_void_p()
        if lib.amdspeech_comm_init(ident, rank, world, C.byref(comm)) != 0:
            return None, lib.amdspeech_last_error().decode("utf-8", "replace")
        return comm, None

    @property
    def device_channel(self):
        """Which library carries the gradient all-reduce of CUDA tensors: "c-abi-rccl" (amdspeech_allreduce_sum_f32),
        "torch-nccl" / "torch-gloo" (torch.distributed fallback), or "none" (one rank)."""
        if self.world == 1:
            return "none"
        if self._comm is not None:
            return "c-abi-rccl"
        import torch.distributed as dist
        return "torch-" + str(dist.get_backend())

    def comm_info(self):
        """{rank, world, rccl_version, lib_path} as the RCCL communicator itself reports them (None without one)."""
        if self._comm is None:
            return None
        r, w, v = C.c_int(), C.c_int(), C.c_int()
        path = C.create_string_buffer(512)
        _l.check(_l.load().amdspeech_comm_info(self._comm, C.byref(r), C.byref(w), C.byref(v), path, 512), "comm_info")
        return {"rank": r.value, "world": w.value, "rccl_version": v.value, "lib_path": path.value.decode()}

    def close(self):
        if self._comm is not None:
            _l.load().amdspeech_comm_destroy(self._comm)
            self._comm = None

    # ---- device channel ------------------------------------------------------------------------------------
    def all_reduce_sum_(self, flat):
        """In-place SUM over ranks of a contiguous float32 buffer (the flat gradient)."""
        if self.world == 1:
            return flat
        assert flat.dtype == torch.float32 and flat.is_contiguous()
        if self._comm is not None and flat.is_cuda:
            stream = C.c_void_p(torch.cuda.current_stream(flat.device).cuda_stream)
            _l.check(_l.load().amdspeech_allreduce_sum_f32(self._comm, stream, C.c_void_p(flat.data_ptr()),
                                                           flat.numel()), "allreduce_sum_f32")
        else:
            import torch.distributed as dist
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        return flat

    def broadcast_(self, flat, root=0):
        if self.world == 1:
            return flat
        assert flat.dtype == torch.float32 and flat.is_contiguous()
        if self._comm is not None and flat.is_cuda:
            stream = C.c_void_p(torch.cuda.current_stream(flat.device).cuda_stream)
            _l.check(_l.load().amdspeech_broadcast_f32(self._comm, stream, C.c_void_p(flat.data_ptr()), flat.numel(),
                                                       int(root)), "broadcast_f32")
        else:
            import torch.distributed as dist
            dist.broadcast(flat, src=root)
        return flat

    # ---- host channel ----------------------------------------------------------------------------------------
    def sum_scalars(self, values):
        """Element-wise SUM over ranks of a short list of python floats (float64 on the wire)."""
        if self.world == 1:
            return [float(v) for v in values]
        import torch.distributed as dist
        t = torch.tensor([float(v) for v in values], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.host_group)
        return [float(v) for v in t]

    def all_true(self, flag):
        """True iff `flag` is true on EVERY rank."""
        if self.world == 1:
            return bool(flag)
        import torch.distributed as dist
        t = torch.tensor([1 if flag else 0], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.host_group)
        return bool(int(t[0]))

    def broadcast_object(self, obj, root=0):
        if self.world == 1:
            return obj
        import torch.distributed as dist
        box = [obj if self.rank == root else None]
        dist.broadcast_object_list(box, src=root, group=self.host_group)
        return box[0]

    def barrier(self):
        if self.world > 1:
            import torch.distributed as dist
            dist.barrier(group=self.host_group)


def current():
    """The job's communicator: joined lazily from the environment on first use."""
    global _group
    if _group is None:
        _group = Group.from_env()
    return _group


def set_current(group):
    global _group
    _group = group
    return group


def shard(items, rank, world):
    """Rank `rank`'s utterances: every world-th item, padded by wrapping around so that EVERY rank holds the same
    number of items -- hence the same number of mini-batches per epoch, hence the same number of collectives."""
    if world <= 1:
        return list(items)
    per = (len(items) + world - 1) // world
    if per == 0:
        return []
    return [items[(rank + i * world) % len(items)] for i in range(per)]


def shard_bucketed(items, batch_size, rank, world, seed=0):
    """Length-bucketed batching for the WHOLE job ("with bucketing, shard within a bucket so ranks see similar T").
    items: [audio, label, duration] lists.  They are sorted by duration, cut into GLOBAL buckets of batch_size * world
    utterances, every bucket is dealt to the ranks like a hand of cards (rank r takes items r, r + world, ... of the sorted
    bucket: batch_size utterances whose longest is the bucket's (r+1)-th longest from the top), and the buckets -- not the
    utterances -- are shuffled with the job-wide `seed`, identically on every rank.  Optimiser step k is then the same global
    bucket everywhere: no rank runs a 10-second mini-batch while the others wait in the all-reduce behind a 3-second one
    (a rank-local bucketed_order does exactly that).  A short last bucket is padded by wrapping around inside itself so that
    every rank holds the same number of utterances (equal shard sizes: the same number of collectives), and stays last.
    The reference has no counterpart (one device; its size ordering: models/SpeechRecognizer.py:58-99)."""
    import random
    world = max(1, int(world))
    ordered = sorted(items, key=lambda it: (it[2] if len(it) > 2 and it[2] is not None else 0.0))
    per = batch_size * world
    groups = [ordered[i:i + per] for i in range(0, len(ordered), per)]
    tail = None
    if groups and len(groups[-1]) < per:
        tail = groups.pop()
        n = len(tail)
        while len(tail) % world:                      # equal hands: repeat the bucket's own utterances
            tail.append(tail[(len(tail) - n) % n])
    random.Random(seed).shuffle(groups)
    if tail:
        groups.append(tail)
    return [it for g in groups for it in g[rank::world]]


def reshuffle_buckets(items, batch_size, seed):
    """A rank's shard_bucketed() list at the end of an epoch: the SAME permutation of the buckets on every rank (same `seed`),
    the utterances of a bucket stay together, a short last bucket stays last."""
    import random
    groups = [items[i:i + batch_size] for i in range(0, len(items), batch_size)]
    tail = [groups.pop()] if groups and len(groups[-1]) < batch_size else []
    random.Random(seed).shuffle(groups)
    return [it for g in groups + tail for it in g]
