"""Row-sharded embedding tables across the GPUs of one node (BASELINE config 4).

The reference keeps ONE concatenated table on one device (src/models/embeddings/base.py:52-57)
and has no distributed code; this is the MI355X-native extension of the same DeepFM path
(src/models/deepfm.py:88-102):

  * one process per GPU, `torch.distributed` backend "nccl" (= RCCL over xGMI);
  * the concatenated [N, D] table and the [N] first-order table are sharded by
    `owner = row % world`, `local_row = row // world`: every field's ids — hot Zipf heads
    included — spread evenly, so lookups per rank balance without knowing the field sizes.
    Every shard ends in one extra SINK row (zero, zero gradients) that padding slots address;
  * data-parallel batch: each rank owns B samples.  A step is

        route (HIP)        rows = x + offsets, stable bucketing by owner into [world, cap] slots
        all-to-all #1      owner-local row ids                       (8 B per slot)
        pack  (HIP)        W[row] and w1[row] into rows of D+4 floats
        all-to-all #2      packed rows back to the requesters        (4(D+4) B per slot)
        slot gather + FM + first-order (one HIP kernel) + the MLP tail, data-parallel
        -- backward --
        FM/gather backward writes gradient rows straight into the outgoing [world*cap, D+4] buffer
        all-to-all #3      gradient rows to the owners -> row-form (COO) gradients of the shards
        ONE flat all-reduce of the replicated tail's gradients (~2 MB)

    xGMI is a point-to-point mesh: an all-to-all is one direct hop per peer pair, no ring.
  * static shapes: every rank sends exactly `cap` slots to every peer, so there is no host sync
    and the compute between the collectives replays as hipGraphs (`enable_graphs`).  A bucket that
    overflows (pathological skew) drops the excess lookups to zeros and raises the sticky flag
    `check_overflow()` reports; `bucket_slack=world` makes overflow impossible (cap = n).

The three device steps come from `ops` (default: the HIP library).  The world-size-2 gloo tests
run on CPU, where the library cannot, and inject torch restatements of the same three steps so
that the choreography — splits, buffer layouts, gradient averaging — is what they exercise.
"""
from typing import List, Optional

import torch
import torch.distributed as dist
from torch import nn

from . import _kernels
from .mlp import run_tail


class HipOps:
    """The product's device steps (csrc/route.hip, csrc/gather_fm.hip, csrc/mlp.hip)."""

    route_buckets = staticmethod(_kernels.route_buckets)
    gather_pack_rows = staticmethod(_kernels.gather_pack_rows)
    unpack_rows = staticmethod(_kernels.unpack_rows)
    slot_fm = staticmethod(_kernels.slot_fm)

    @staticmethod
    def tail(deep_branch: nn.Sequential, x: torch.Tensor, y_fm: torch.Tensor, labels=None, loss_seed=None) -> torch.Tensor:
        return run_tail(deep_branch, x, last_add=y_fm, labels=labels, loss_seed=loss_seed).squeeze(-1)


def bucket_capacity(n: int, world: int, slack: float) -> int:
    """Slots per peer bucket: the mean load with `slack` headroom plus six standard deviations."""
    if world == 1:
        return n
    mean = -(-n // world)
    return min(n, int(mean * slack + 6.0 * (mean ** 0.5)) + 64)


_captures = 0          # global-mode captures this process has under way (noted by note_capture())


def note_capture(delta: int) -> None:
    """Callers that capture a hipGraph while a DirectComm may exist bracket the capture with note_capture(+1) / (-1)."""
    global _captures
    _captures = max(0, _captures + delta)


def _capture_under_way() -> bool:
    return _captures > 0


class _relaxed_capture_mode:
    """Thread-local relaxed capture mode for the calling thread (cudaThreadExchangeStreamCaptureMode): its event queries
    neither join nor invalidate a capture another thread holds.  A no-op where torch does not expose the call."""

    def __enter__(self):
        self._rt = self._old = None
        try:
            rt = torch.cuda.cudart()
            self._old = rt.cudaThreadExchangeStreamCaptureMode(rt.cudaStreamCaptureMode.Relaxed) if hasattr(
                rt, "cudaThreadExchangeStreamCaptureMode") else None
            self._rt = rt
        except Exception:  # noqa: BLE001
            self._rt = None
        return self

    def __exit__(self, *exc):
        if self._rt is not None and self._old is not None:
            try:
                self._rt.cudaThreadExchangeStreamCaptureMode(self._old)
            except Exception:  # noqa: BLE001
                pass
        return False


class DirectComm:
    """The library's own RCCL communicator (csrc/comm.hip): all-to-all / all-reduce enqueued on the CURRENT stream, i.e.
    the one the kernels run on — no process-group stream, no event hand-offs around the collectives (≈60 us of the
    one-rank sharded step).  Built collectively over `group` (an NCCL/RCCL group with one rank per GPU);
    `create` returns None — and the caller keeps torch.distributed's collectives — when the group is not on RCCL, the
    switch MI_DIRECT_RCCL=0 is set, or the communicator cannot be made."""

    def __init__(self, handle, world):
        self.handle, self.world = handle, world
        self._pending = []          # (host time, event) markers behind collectives, oldest first
        self._last_mark = 0.0
        self._watchdog = None
        import threading
        self._lock = threading.Lock()      # _pending and handle are touched by the watchdog thread too

    # A collective of a library-owned communicator has no ProcessGroupNCCL watchdog behind it: when a peer dies (or
    # raises and leaves), the surviving ranks would sit in the next all-to-all forever.  Every second or so a collective
    # leaves an event behind; a daemon thread gives the oldest unfinished one MI_COMM_DEADLINE_S seconds (default 180),
    # then aborts the communicator and EXITS the process with status 86 (never re-execs: the GPU-box rules) so that the
    # launcher tears the job down.
    def _mark(self, device):
        import os
        import threading
        import time

        now = time.monotonic()
        if now - self._last_mark < 1.0:
            return
        self._last_mark = now
        if torch.cuda.is_current_stream_capturing():
            return                      # an event recorded into a capture cannot be queried from outside it
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(device))
        with self._lock:
            self._pending.append((now, ev))
        if self._watchdog is None:
            deadline = float(os.environ.get("MI_COMM_DEADLINE_S", "180"))

            def watch():
                import sys

                from . import _lib
                while self.handle is not None:
                    time.sleep(1.0)
                    # A cross-thread event query while ANOTHER thread holds a global-mode stream capture (trainer's
                    # GraphedTrainStep, bench's per-batch graphs) would invalidate that capture: this thread's own
                    # calls are made capture-"relaxed" (what ProcessGroupNCCL's watchdog does), and a tick that finds a
                    # capture under way on this device is skipped besides — captures last milliseconds, the deadline is
                    # minutes.
                    late = False
                    with self._lock:
                        if self.handle is None:
                            return
                        if _capture_under_way():
                            continue
                        with _relaxed_capture_mode():
                            while self._pending and self._pending[0][1].query():
                                self._pending.pop(0)
                        late = bool(self._pending) and time.monotonic() - self._pending[0][0] > deadline
                        handle = self.handle
                    if late:
                        print(f"mi355x_recsys: a collective of the direct RCCL communicator has not finished within "
                              f"{deadline:.0f} s (a peer is gone?); aborting the communicator and exiting", file=sys.stderr, flush=True)
                        try:
                            _lib.load().mi_comm_abort(handle)
                        finally:
                            os._exit(86)

            self._watchdog = threading.Thread(target=watch, name="mi-comm-watchdog", daemon=True)
            self._watchdog.start()

    @staticmethod
    def create(group, device):
        import ctypes
        import os
        import warnings

        from . import _lib

        if os.environ.get("MI_DIRECT_RCCL", "1") == "0" or dist.get_backend(group) != "nccl" or device is None:
            return None
        if torch.device(device).type != "cuda":
            return None
        lib = _lib.load()
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        # every rank must be able to load RCCL itself BEFORE anyone enters mi_comm_init (which waits for all ranks)
        can = torch.tensor([int(lib.mi_comm_available() == 0)], device=device)
        dist.all_reduce(can, op=dist.ReduceOp.MIN, group=group)
        if int(can.item()) == 0:
            warnings.warn("mi355x_recsys: RCCL cannot be loaded on every rank; using torch.distributed's collectives")
            return None
        buf = ctypes.create_string_buffer(128)
        ok = 1
        if rank == 0:
            ok = int(lib.mi_comm_unique_id(buf) == 0)
        box = [bytes(buf.raw) if ok else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group else 0, group=group)
        if box[0] is None:
            warnings.warn("mi355x_recsys: no RCCL for the direct communicator; using torch.distributed's collectives")
            return None
        handle = ctypes.c_void_p()
        with torch.cuda.device(device):
            rc = lib.mi_comm_init(box[0], world, rank, ctypes.byref(handle))
        comm = DirectComm(handle, world) if rc == 0 else None
        ok = comm is not None and comm._self_test(rank, device)
        flag = torch.tensor([int(ok)], device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)          # all ranks or none
        if int(flag.item()) == 0:
            if comm is not None and comm.handle is not None:
                comm.close()
            warnings.warn("mi355x_recsys: the direct RCCL communicator did not come up on every rank; using torch.distributed")
            return None
        return comm

    def _self_test(self, rank: int, device, deadline_s: float = 60.0) -> bool:
        """One all-to-all and one all-reduce with known answers on a side stream, waited for with a deadline: a
        communicator whose first collectives do not finish (or deliver the wrong peer's bytes) is aborted here, at
        start-up, rather than hanging the first training step."""
        import time

        from . import _lib

        w = self.world
        side = torch.cuda.Stream(device=device)
        done = torch.cuda.Event()
        with torch.cuda.device(device), torch.cuda.stream(side):
            send = (torch.arange(w, device=device, dtype=torch.int32) + rank * w).repeat_interleave(16).contiguous()
            recv = torch.full_like(send, -1)
            ones = torch.ones(256, dtype=torch.float32, device=device)
            try:
                self.all_to_all(recv, send)
                self.all_reduce_sum(ones)
            except RuntimeError:
                self.close()
                return False
            done.record(side)
        t_end = time.monotonic() + deadline_s
        while not done.query():
            if time.monotonic() > t_end:
                _lib.load().mi_comm_abort(self.handle)
                self.handle = None
                return False
            time.sleep(0.002)
        want = (torch.arange(w, device=device, dtype=torch.int32) * w + rank).repeat_interleave(16)
        good = bool(torch.equal(recv, want)) and bool((ones == float(w)).all())
        if not good:
            self.close()
        return good

    def all_to_all(self, out: torch.Tensor, inp: torch.Tensor):
        from . import _lib

        assert out.is_contiguous() and inp.is_contiguous() and out.numel() == inp.numel() and out.dtype == inp.dtype
        nbytes = inp.numel() * inp.element_size()
        assert nbytes % self.world == 0
        _lib.check(_lib.load().mi_comm_all_to_all(self.handle, inp.data_ptr(), out.data_ptr(), nbytes // self.world,
                                                  _lib.stream_ptr(inp.device)), "mi_comm_all_to_all")
        self._mark(inp.device)

    def all_reduce_sum(self, flat: torch.Tensor):
        from . import _lib

        assert flat.is_contiguous() and flat.dtype == torch.float32
        _lib.check(_lib.load().mi_comm_all_reduce_sum_f32(self.handle, flat.data_ptr(), flat.numel(), _lib.stream_ptr(flat.device)),
                   "mi_comm_all_reduce_sum_f32")
        self._mark(flat.device)

    def close(self):
        from . import _lib

        with self._lock:                 # (not while the watchdog is between reading the handle and aborting it)
            handle, self.handle = self.handle, None
        if handle is not None:
            _lib.load().mi_comm_destroy(handle)


def _all_to_all(model, out, inp):
    """Equal-split all-to-all of the sharded lookup: on the compute stream through the library's communicator when there
    is one, else torch.distributed's."""
    comm = model.__dict__.get("_comm")
    if comm is not None:
        comm.all_to_all(out, inp)
    else:
        dist.all_to_all_single(out, inp, group=model.group)


def expected_peak_load(field_dims: List[int], batch: int, world: int) -> float:
    """Largest expected number of lookups one owner receives from a batch of `batch` samples when the ids of every field
    are uniform over that field: owner = (offset_f + id) % world, so a field with fewer values than ranks sends ALL its
    `batch` lookups to at most `cardinality` owners (a 3-value Criteo field: batch/3 each to 3 owners, nothing to the
    other 5) — the plain mean n/world underestimates that."""
    load = [0.0] * world
    off = 0
    for c in field_dims:
        base, extra = divmod(c, world)               # owners of rows off .. off+c-1
        for o in range(world):
            rows = base + (1 if ((o - off) % world) < extra else 0)
            load[o] += batch * rows / c
        off += c
    return max(load)


def field_bucket_capacity(field_dims: List[int], batch: int, world: int, slack: float) -> int:
    """bucket_capacity sized for the fields at hand: expected_peak_load with `slack` headroom + six standard deviations."""
    n = batch * len(field_dims)
    if world == 1:
        return n
    peak = expected_peak_load(field_dims, batch, world)
    return min(n, int(peak * slack + 6.0 * (peak ** 0.5)) + 64)


class _Exchange(torch.autograd.Function):
    """x[B,F] raw ids -> (packed rows received [world*cap + 1, D+4], slot[B,F]).

    Backward ships the gradient of the received buffer to the owners and returns it as
    uncoalesced COO gradients of the LOCAL shards (what nn.Embedding(sparse=True) gives a local
    table), averaged over ranks like the dense tail's all-reduce (every rank's loss is a mean over
    ITS samples; the sum over ranks / world is the gradient of the global-batch mean)."""

    @staticmethod
    def forward(ctx, x, W_local, w1_local, model):
        ops, world, group = model.ops, model.world, model.group
        D = W_local.shape[1]
        cap = model.capacity(x.shape[0])
        S = world * cap
        static = model.__dict__.get("_static_io")
        use_static = static is not None and static[0].shape[0] == S + 1 and tuple(static[1].shape) == tuple(x.shape)
        send_rows, slot = ops.route_buckets(x, model.offsets, world, model.num_rows, cap, model.bucket_overflow,
                                            slot_out=static[1] if use_static else None)
        local_rows = torch.empty_like(send_rows)
        _all_to_all(model, local_rows, send_rows)
        packed = ops.gather_pack_rows(local_rows, W_local, w1_local)              # [S, D+4]
        if use_static:
            recv = static[0].detach()        # same storage (no copy into the graph), fresh autograd identity
        else:
            recv = packed.new_empty((S + 1, D + 4))
            recv[S].zero_()                  # the dump slot reads as a zero row
        _all_to_all(model, recv[:S], packed)
        ctx.model = model
        ctx.local_rows, ctx.meta = local_rows, (S, D, world, group, tuple(W_local.shape), tuple(w1_local.shape))
        ctx.mark_non_differentiable(slot)
        return recv, slot

    @staticmethod
    def backward(ctx, g_recv, _g_slot):
        S, D, world, group, Wshape, w1shape = ctx.meta
        g_owner = g_recv.new_empty((S, D + 4))
        _all_to_all(ctx.model, g_owner, g_recv[:S].contiguous())
        inv = 1.0 / world
        idx = ctx.local_rows.view(1, -1)
        gW = gw1 = None
        if ctx.needs_input_grad[1]:
            gW = torch.sparse_coo_tensor(idx, g_owner[:, :D] * inv, Wshape, check_invariants=False)
        if ctx.needs_input_grad[2]:
            g1 = (g_owner[:, D] * inv).view((-1,) + (1,) * (len(w1shape) - 1))
            gw1 = torch.sparse_coo_tensor(idx, g1, w1shape, check_invariants=False)
        return None, gW, gw1, None


def shard_rows(full: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """The rows of a full [N, ...] table that rank owns (row % world == rank), in local order."""
    return full[rank::world].contiguous()


def local_num_rows(N: int, rank: int, world: int) -> int:
    return (N - rank + world - 1) // world


class ShardedDeepFM(nn.Module):
    """DeepFM (src/models/deepfm.py:11-105) with row-sharded tables and a replicated MLP tail.

    Constructor arguments as the reference's DeepFM where they apply.  `embedding_shard`
    ([N_local + 1, D]) and `fc_shard` ([N_local + 1, 1]) hold this rank's rows plus the sink row; use
    `load_full_tables` to shard a reference checkpoint's `embedding._emb_module.weight` /
    `fc.weight`.  Table gradients are row-form (sparse COO) on the local shards.
    """

    def __init__(self, field_dims: List[int], num_factor: int, hidden_sizes: List[int], p_dropout: float = 0.1,
                 use_batchnorm=False, device=None, process_group=None, ops=None, bucket_slack: float = 1.25):
        super().__init__()
        self.ops = ops or HipOps
        self.bucket_slack = bucket_slack
        self.field_dims = list(field_dims)
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.rank = dist.get_rank(process_group)
        N = sum(field_dims)
        self.num_rows = N
        self.num_local_rows = local_num_rows(N, self.rank, self.world)
        # xavier-uniform over the GLOBAL [N, D] matrix (src/models/embeddings/base.py:66-67); sink row zero
        bound = (6.0 / (N + num_factor)) ** 0.5
        # (in place: a 1e9-row table is 64 GB, temporaries of that size must not pile up)
        W = torch.empty(self.num_local_rows + 1, num_factor, device=device).uniform_(-bound, bound)
        w1 = torch.empty(self.num_local_rows + 1, 1, device=device).normal_()      # N(0,1) like nn.EmbeddingBag
        W[-1].zero_()
        w1[-1].zero_()
        self.embedding_shard = nn.Parameter(W)
        self.fc_shard = nn.Parameter(w1)
        self._bias = nn.Parameter(torch.zeros(1, device=device))
        deep_in = num_factor * len(field_dims)
        layers: List[nn.Module] = []
        for size in hidden_sizes:
            layers.append(nn.Linear(deep_in, size))
            if use_batchnorm:
                layers.append(nn.BatchNorm1d(size))
            layers.append(nn.ReLU())
            layers.append(nn.Dropout(p_dropout))
            deep_in = size
        layers.append(nn.Linear(deep_in, 1))
        self._deep_branch = nn.Sequential(*layers).to(device)
        fd = torch.cat([torch.tensor([0], dtype=torch.long), torch.tensor(field_dims)])
        self.register_buffer("offsets", torch.cumsum(fd[:-1], 0).unsqueeze(0).to(device))
        self.register_buffer("bucket_overflow", torch.zeros(1, dtype=torch.int32, device=device), persistent=False)
        self.sync_dense_parameters()
        # the library's own communicator (collectives on the compute stream) when the group runs on RCCL
        self.__dict__["_comm"] = DirectComm.create(process_group, device) if ops is None else None

    # ---- parameter plumbing ------------------------------------------------------------
    def dense_parameters(self):
        return [self._bias] + list(self._deep_branch.parameters())

    def sync_dense_parameters(self):
        """Replicas start from rank 0's dense weights."""
        for p in self.dense_parameters():
            dist.broadcast(p.data, src=dist.get_global_rank(self.group, 0) if self.group else 0, group=self.group)

    def allreduce_dense_grads(self):
        """Average the replicated tail's gradients with one flat collective."""
        grads = [p.grad for p in self.dense_parameters() if p.grad is not None]
        if not grads:
            return
        flat = torch.cat([g.reshape(-1) for g in grads])
        if self.__dict__.get("_comm") is not None:
            self._comm.all_reduce_sum(flat)
            flat /= self.world
        elif dist.get_backend(self.group) == "nccl":
            dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=self.group)
        else:
            dist.all_reduce(flat, group=self.group)
            flat /= self.world
        torch._foreach_copy_(grads, [v.view_as(g) for v, g in zip(flat.split([g.numel() for g in grads]), grads)])

    def get_optimizers(self, config):
        """The optimizers `get_optimizers` (src/models/deepfm.py:155-219) builds for its `sparse: True` configs, for this
        model: the row-sparse step on the two table shards (their gradients arrive in row form, already averaged over
        the global batch), the dense step on the replicated tail (whose gradients are all-reduced, so every rank takes
        the same step and the replicas stay identical)."""
        from . import optim

        name = config.get("optimizer", "adam")
        lr_emb = config.get("learning_rate_emb", config["learning_rate"])
        shards = [self.embedding_shard, self.fc_shard]
        if name == "adam":
            return [optim.SparseAdam(shards, lr=lr_emb, capturable=True),
                    optim.Adam(self.dense_parameters(), lr=config["learning_rate"], weight_decay=config["weight_decay"])]
        if name == "sgd":
            return [optim.SparseSGD(shards, lr=lr_emb),
                    torch.optim.SGD(self.dense_parameters(), lr=config["learning_rate"], weight_decay=config["weight_decay"])]
        raise ValueError(f"optimizer_name={name!r} is not recognized")

    def capacity(self, batch: int) -> int:
        """Slots per peer bucket for a batch of `batch` samples (every rank computes the same number)."""
        return field_bucket_capacity(self.field_dims, batch, self.world, self.bucket_slack)

    def _collective_flags(self) -> List[int]:
        """[bucket overflow, index-error bits] OR-ed (MAX, bit by bit) over the ranks: a rank that raised alone would leave
        the others waiting in the next all-to-all, so every check is a collective and every rank raises together."""
        from . import _lib

        dev = self.embedding_shard.device
        # (the error word lives with the HIP kernels; the CPU choreography tests inject torch restatements and have none)
        err = _lib.err_word(dev).view(-1)[0] if dev.type == "cuda" else torch.zeros((), dtype=torch.int32)
        flags = torch.stack([self.bucket_overflow.view(-1)[0].to(torch.int32), (err & 1).to(torch.int32),
                             ((err >> 1) & 1).to(torch.int32)])
        if self.world > 1:
            dist.all_reduce(flags, op=dist.ReduceOp.MAX, group=self.group)
        return [int(v) for v in flags.tolist()]

    def check_overflow(self):
        """Synchronise; raise ON EVERY RANK if a fixed-capacity bucket overflowed on any rank since the last check."""
        if self._collective_flags()[0]:
            self.bucket_overflow.zero_()
            raise RuntimeError("sharded lookup: a peer bucket overflowed its fixed capacity on some rank (extreme id skew); "
                               "raise bucket_slack (bucket_slack=world can never overflow)")

    def check_index_errors(self):
        """_lib.check_index_errors() for the sharded model: the ranks agree on the error bits first, then all raise."""
        from . import _lib

        _, oob, off_field = self._collective_flags()
        if oob or off_field:
            if self.embedding_shard.device.type == "cuda":
                _lib.err_word(self.embedding_shard.device).zero_()
            if oob:
                raise IndexError("index out of range in embedding lookup on some rank (mi355x_recsys)")
            raise IndexError("an id lies outside its own field's range on some rank (mi355x_recsys)")

    @torch.no_grad()
    def load_full_tables(self, embedding_weight: torch.Tensor, fc_weight: torch.Tensor):
        n = self.num_local_rows
        self.embedding_shard[:n].copy_(shard_rows(embedding_weight, self.rank, self.world))
        self.fc_shard[:n].copy_(shard_rows(fc_weight, self.rank, self.world))
        self.embedding_shard[n].zero_()
        self.fc_shard[n].zero_()

    # ---- the compute between the collectives ------------------------------------------------
    def _local_compute(self, recv, slot, labels=None, loss_seed=None):
        """received packed rows + slots -> logits [B]: slot gather, FM, first-order term, MLP tail.
        labels / loss_seed (the graphed step): the targets and the scalar its backward is seeded with (1 / world) — the
        tail's head launch then evaluates the BCE criterion and the head's backward as well (mlp.run_tail)."""
        fused = self._fused_local(recv, slot, labels, loss_seed)
        if fused is not None:
            return fused
        emb, y_fm = self.ops.slot_fm(recv, slot, self._bias)
        if labels is None:
            return self.ops.tail(self._deep_branch, emb.reshape(slot.shape[0], -1), y_fm)
        return self.ops.tail(self._deep_branch, emb.reshape(slot.shape[0], -1), y_fm, labels, loss_seed)

    def _fused_local(self, recv, slot, labels, loss_seed):
        """The local compute as ONE autograd node (tail.SlotDeepFMFusedFn) when the fused tail takes the MLP: the slot lookup
        carries the tail's dropout bits and zero fill, the lookup's backward runs in the epilogue of the tail's first
        input-gradient product and writes straight into the receive buffer's gradient.  None: the two-node path."""
        from . import mlp as _mlp, tail as _tail

        D = recv.shape[1] - 4
        if not (self.ops is HipOps and _mlp.FUSED_TAIL and _tail.FM_EPILOGUE and torch.is_grad_enabled() and recv.is_cuda
                and recv.requires_grad and slot.dim() == 2 and D >= 4 and D % 4 == 0 and (D // 4) & (D // 4 - 1) == 0 and D <= 256):
            return None
        groups = _mlp._groups(self._deep_branch)
        plan = _tail.fused_tail_plan(self._deep_branch, _tail._InputSpec(slot.shape[0], slot.shape[1] * D, recv.device), groups)
        if plan is None:
            return None
        return _tail.run_fused_slot_deepfm(plan, groups[-1][1], _mlp._seed_word(recv.device), recv, slot, self._bias,
                                           labels if self.training else None, loss_seed).squeeze(-1)

    def enable_graphs(self, batch_size: int):
        """Capture the local compute (forward AND backward) for a fixed batch size as hipGraphs
        (torch.cuda.make_graphed_callables).  The RCCL collectives stay outside the graphs —
        recording them hung on this stack — so a step is: routing + two all-to-alls, ONE graph replay
        for gather+FM+MLP forward, one for their backward, the gradient all-to-all + all-reduce.
        The exchange writes straight into the graphs' static inputs (no staging copy)."""
        F, D = self.offsets.shape[1], self.embedding_shard.shape[1]
        dev = self.embedding_shard.device
        S = self.world * self.capacity(batch_size)
        # no collective may be in flight while a capture is open (the RCCL watchdog polls events)
        torch.cuda.synchronize(dev)
        dist.barrier(group=self.group)
        torch.cuda.synchronize(dev)
        recv = torch.zeros(S + 1, D + 4, device=dev).requires_grad_(True)
        slot = torch.arange(batch_size * F, device=dev).view(batch_size, F) % max(S, 1)

        class _Local(nn.Module):
            def __init__(inner, outer):
                super().__init__()
                inner.outer = [outer]            # not registered as a submodule (no param duplication)
                inner._bias = outer._bias
                inner._deep_branch = outer._deep_branch

            def forward(inner, r, s):
                return inner.outer[0]._local_compute(r, s)

        # kept out of the module registry (it shares this module's parameters; state_dict must not change)
        object.__setattr__(self, "_graphed_local",
                           torch.cuda.make_graphed_callables(_Local(self), (recv, slot), num_warmup_iters=3))
        object.__setattr__(self, "_graphed_batch", batch_size)
        with torch.no_grad():
            recv.zero_()
        object.__setattr__(self, "_static_io", (recv, slot))
        torch.cuda.synchronize(dev)
        dist.barrier(group=self.group)

    def make_graphed_step(self, criterion, batch_size: int, static_labels: Optional[torch.Tensor] = None):
        """A whole training step `step(x, y) -> loss` for a fixed batch size with everything between the
        collectives — slot gather + FM + MLP forward, the criterion, and the complete backward down to the
        outgoing gradient rows and one FLAT buffer of the replicated tail's gradients — replayed as ONE
        hipGraph.  Outside the graph a step is four library launches (routing, packing), three all-to-alls,
        two strided scales that form the shards' COO gradients, and the flat all-reduce: no autograd
        bookkeeping, no host sync.  After a step `p.grad` of every parameter is set exactly as by
        `criterion(model(x), y).backward(); model.allreduce_dense_grads()` (dense gradients are views of
        the flat buffer); the returned loss is a device scalar that the next step overwrites.
        static_labels ([batch_size] float, optional): a buffer the caller refreshes itself before every step
        (then `step(x)` needs no label copy)."""
        F, D = self.offsets.shape[1], self.embedding_shard.shape[1]
        dev = self.embedding_shard.device
        world, group = self.world, self.group
        cap = self.capacity(batch_size)
        S = world * cap
        dense = self.dense_parameters()
        recv = torch.zeros(S + 1, D + 4, device=dev).requires_grad_(True)
        slot = (torch.arange(batch_size * F, device=dev) % max(S, 1)).view(batch_size, F)
        ys = static_labels if static_labels is not None else torch.zeros(batch_size, device=dev)
        # d(global-batch mean loss)/d(this rank's mean loss) = 1/world: seeded into the backward so that the table
        # gradients arrive already averaged and the dense ones only need a SUM all-reduce
        seed_grad = torch.full((), 1.0 / self.world, device=dev)

        from . import losses as _losses
        in_head = isinstance(criterion, _losses.BCEWithLogitsLoss) and self.ops is HipOps      # (the gloo tests' CPU ops take no labels)

        def local():
            logits = self._local_compute(recv, slot, ys, seed_grad) if in_head else self._local_compute(recv, slot)
            loss = criterion(logits, ys)
            grads = torch.autograd.grad(loss, [recv] + dense, grad_outputs=seed_grad)
            return loss, grads[0], torch.cat([g.reshape(-1) for g in grads[1:]])

        torch.cuda.synchronize(dev)
        dist.barrier(group=group)              # no collective in flight while a capture is open
        torch.cuda.synchronize(dev)
        try:
            side = torch.cuda.Stream(dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for _ in range(3):
                    local()
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            graph = torch.cuda.CUDAGraph()
            # thread_local: the RCCL watchdog thread may poll its events while this thread records
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                loss, g_recv, flat = local()
        finally:                               # every rank reaches the closing barrier, failed capture or not
            torch.cuda.synchronize(dev)
            dist.barrier(group=group)

        sizes = [p.numel() for p in dense]
        views = [v.view_as(p) for v, p in zip(flat.split(sizes), dense)]
        Wshape, w1shape = tuple(self.embedding_shard.shape), tuple(self.fc_shard.shape)

        def step(x, y=None, mark=None):
            """mark (optional callable): called with a phase name after each phase has been ENQUEUED (bench.py records a
            timing event there; the default path calls nothing)."""
            if tuple(x.shape) != (batch_size, F):
                raise ValueError(f"this step was captured for x of shape {(batch_size, F)}, got {tuple(x.shape)}")
            if y is not None and y.data_ptr() != ys.data_ptr():
                ys.copy_(y)
            send_rows, _ = self.ops.route_buckets(x, self.offsets, world, self.num_rows, cap, self.bucket_overflow,
                                                  slot_out=slot)
            local_rows = torch.empty_like(send_rows)
            if mark: mark("route (2 launches)")
            _all_to_all(self, local_rows, send_rows)
            if mark: mark("all-to-all #1: row ids")
            packed = self.ops.gather_pack_rows(local_rows, self.embedding_shard, self.fc_shard)
            if mark: mark("pack rows at the owner")
            with torch.no_grad():
                _all_to_all(self, recv[:S], packed)
                if mark: mark("all-to-all #2: packed rows")
                graph.replay()
                if mark: mark("local graph: slot gather + FM + MLP fwd, loss, whole bwd")
                g_owner = torch.empty_like(packed)
                _all_to_all(self, g_owner, g_recv[:S])
                if mark: mark("all-to-all #3: gradient rows")
                idx = local_rows.view(1, -1)
                # the received rows are already scaled by 1/world; COO values must be contiguous (torch's sparse
                # kernels read strided values wrongly), so the two column blocks are copied out (one launch)
                gvals, glin = self.ops.unpack_rows(g_owner, D)
                self.embedding_shard.grad = torch.sparse_coo_tensor(idx, gvals, Wshape, check_invariants=False)
                self.fc_shard.grad = torch.sparse_coo_tensor(idx, glin.view((-1,) + tuple(w1shape[1:])), w1shape,
                                                             check_invariants=False)
                if mark: mark("COO gradient column copies")
                if self.__dict__.get("_comm") is not None:
                    self._comm.all_reduce_sum(flat)
                else:
                    dist.all_reduce(flat, group=group)
                if mark: mark("flat all-reduce of the dense gradients")
            for p, v in zip(dense, views):
                p.grad = v
            return loss

        step.keepalive = (seed_grad,)      # read by every replay: must outlive this function's frame
        return step

    def forward(self, x):
        """x: int [B_local, F] raw per-field ids -> logits [B_local]."""
        recv, slot = _Exchange.apply(x, self.embedding_shard, self.fc_shard, self)
        graphed = self.__dict__.get("_graphed_local")
        if graphed is not None and x.shape[0] == self._graphed_batch and self.training:
            return graphed(recv, slot)
        return self._local_compute(recv, slot)
