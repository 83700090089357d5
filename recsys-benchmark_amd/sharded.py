"""Row-sharded embedding tables across the GPUs of one node (BASELINE config 4).

The reference keeps ONE concatenated table on one device (src/models/embeddings/base.py:52-57)
and has no distributed code; this is the MI355X-native extension of the same DeepFM path:

  * one process per GPU, `torch.distributed` backend "nccl" (= RCCL over xGMI);
  * the concatenated [N, D] table and the [N] first-order table are sharded by
    `owner = row % world`, `local_row = row // world`: every field's ids — hot Zipf heads
    included — spread evenly, so lookups per rank balance without knowing the field sizes;
  * data-parallel batch: each rank owns B samples.  Forward = all-to-all #1 (row ids to their
    owners) -> local HIP row gather -> all-to-all #2 (rows + first-order weights back) -> the
    fused FM / first-order kernel and the MLP on the sample's owner.  Backward mirrors it:
    one all-to-all of gradient rows, which land as row-form (COO) gradients of the local shards.
    xGMI is a point-to-point mesh: all-to-all is one direct hop per peer pair, no ring.
  * the dense tail (MLP, bias) is replicated; its gradients are averaged with ONE flat
    all-reduce (~2 MB, latency-bound).

Only the local lookup touches the HIP library; the routing (bucketing, splits, permutations,
collectives) is plain torch so it is covered on CPU by gloo tests with the lookup injected.
"""
from typing import Callable, List, Optional

import torch
import torch.distributed as dist
from torch import nn

from . import _kernels
from .mlp import run_tail


def _hip_gather(W: torch.Tensor, local_rows: torch.Tensor) -> torch.Tensor:
    return _kernels.gather_rows(local_rows, W)


class _Route:
    """Bucket n global row ids by owner and remember how to undo it."""

    def __init__(self, rows: torch.Tensor, world: int, group):
        flat = rows.reshape(-1)
        owner = flat % world
        # stable sort by owner keeps the (sample, field) order inside each bucket
        self.perm = torch.argsort(owner, stable=True)
        self.send_rows = (flat // world)[self.perm].contiguous()      # owner-local row ids
        send_counts = torch.bincount(owner, minlength=world)
        recv_counts = torch.empty_like(send_counts)
        dist.all_to_all_single(recv_counts, send_counts, group=group)
        # split sizes must be host ints for all_to_all_single: one small sync per step
        both = torch.stack([send_counts, recv_counts]).tolist()
        self.send_splits: List[int] = both[0]
        self.recv_splits: List[int] = both[1]
        self.n_send = flat.numel()
        self.n_recv = sum(self.recv_splits)
        self.group = group

    def to_owner(self, t: torch.Tensor) -> torch.Tensor:
        """t is [n_send, ...] in bucket order -> [n_recv, ...] on the owners."""
        out = t.new_empty((self.n_recv,) + tuple(t.shape[1:]))
        dist.all_to_all_single(out, t.contiguous(), self.recv_splits, self.send_splits, group=self.group)
        return out

    def to_requester(self, t: torch.Tensor) -> torch.Tensor:
        """t is [n_recv, ...] on the owners -> [n_send, ...] in bucket order at the requesters."""
        out = t.new_empty((self.n_send,) + tuple(t.shape[1:]))
        dist.all_to_all_single(out, t.contiguous(), self.send_splits, self.recv_splits, group=self.group)
        return out

    def unpermute(self, t: torch.Tensor) -> torch.Tensor:
        out = torch.empty_like(t)
        out[self.perm] = t
        return out


class _FixedRoute:
    """Static-shape routing: every rank sends exactly `cap` slots to every peer (padding slots carry
    local row 0 forward and zero gradients backward), so there is no host sync, no data-dependent
    shape, and the whole step can be captured in a hipGraph.  `cap` = ceil(n/world) * slack + 6 sigma;
    a bucket that still overflows (pathological skew) drops the excess lookups to zeros and raises
    the sticky `overflow` flag that `ShardedDeepFM.check_overflow()` turns into an error."""

    def __init__(self, rows: torch.Tensor, world: int, group, slack: float, overflow: torch.Tensor):
        flat = rows.reshape(-1)
        n = flat.numel()
        mean = -(-n // world)
        cap = n if world == 1 else min(n, int(mean * slack + 6.0 * (mean ** 0.5)) + 64)
        self.cap, self.world, self.group, self.n = cap, world, group, n
        owner = flat % world
        order = torch.argsort(owner, stable=True)
        owner_sorted = owner[order]
        # (torch.bincount reads its max back to the host: not capturable)
        counts = (owner.unsqueeze(1) == torch.arange(world, device=flat.device).unsqueeze(0)).sum(0)
        starts = torch.cumsum(counts, 0) - counts
        pos = torch.arange(n, device=flat.device) - starts[owner_sorted]
        fits = pos < cap
        overflow.logical_or_((~fits).any().view(1))
        dump = world * cap                               # one extra slot swallows the overflow
        slot_sorted = torch.where(fits, owner_sorted * cap + pos, torch.full_like(pos, dump))
        # slot of every ORIGINAL lookup position (dump for dropped ones)
        self.slot = torch.empty_like(slot_sorted)
        self.slot[order] = slot_sorted
        send = torch.zeros(world * cap + 1, dtype=flat.dtype, device=flat.device)
        send[self.slot] = flat // world
        self.send_rows = send[:-1]

    def exchange(self, t: torch.Tensor) -> torch.Tensor:
        out = torch.empty_like(t)
        dist.all_to_all_single(out, t.contiguous(), group=self.group)
        return out

    def scatter_to_slots(self, t: torch.Tensor) -> torch.Tensor:
        """[n, ...] in lookup order -> [world*cap, ...] send buffer (zeros in padding slots)."""
        buf = t.new_zeros((self.world * self.cap + 1,) + tuple(t.shape[1:]))
        buf[self.slot] = t
        return buf[:-1]

    def gather_from_slots(self, buf: torch.Tensor) -> torch.Tensor:
        """[world*cap, ...] returned buffer -> [n, ...] in lookup order (zeros for dropped lookups)."""
        ext = torch.cat([buf, buf.new_zeros((1,) + tuple(buf.shape[1:]))])
        return ext[self.slot]


class ShardedLookupFixed(torch.autograd.Function):
    """Same contract as ShardedLookup with the static-shape routing (the default on GPUs)."""

    @staticmethod
    def forward(ctx, rows, W_local, w1_local, world: int, group, gather: Callable, slack: float, overflow):
        route = _FixedRoute(rows, world, group, slack, overflow)
        local_rows = route.exchange(route.send_rows)                          # [world*cap]
        emb = route.gather_from_slots(route.exchange(gather(W_local, local_rows)))
        lin = route.gather_from_slots(route.exchange(gather(w1_local.view(-1, 1), local_rows))).view(-1)
        ctx.route, ctx.local_rows = route, local_rows
        ctx.shapes = (tuple(W_local.shape), tuple(w1_local.shape))
        return emb, lin

    @staticmethod
    def backward(ctx, g_emb, g_lin):
        route, local_rows = ctx.route, ctx.local_rows
        Wshape, w1shape = ctx.shapes
        inv = 1.0 / route.world
        gW = gw1 = None
        if ctx.needs_input_grad[1]:
            g_rows = route.exchange(route.scatter_to_slots(g_emb.contiguous())).mul_(inv)
            gW = torch.sparse_coo_tensor(local_rows.view(1, -1), g_rows, Wshape, check_invariants=False)
        if ctx.needs_input_grad[2]:
            g1 = route.exchange(route.scatter_to_slots(g_lin.contiguous().view(-1, 1))).mul_(inv)
            gw1 = torch.sparse_coo_tensor(local_rows.view(1, -1), g1.view((-1,) + (1,) * (len(w1shape) - 1)),
                                          w1shape, check_invariants=False)
        return None, gW, gw1, None, None, None, None, None


class ShardedLookup(torch.autograd.Function):
    """(emb[n,D], lin[n]) for n global rows from the row-sharded tables.

    Backward ships the gradient rows to the owners and returns them as uncoalesced COO
    gradients of the LOCAL shards (what nn.Embedding(sparse=True) would give a local table).
    """

    @staticmethod
    def forward(ctx, rows, W_local, w1_local, world: int, group, gather: Callable):
        route = _Route(rows, world, group)
        local_rows = route.to_owner(route.send_rows)
        emb_owner = gather(W_local, local_rows)                     # [n_recv, D]
        lin_owner = gather(w1_local.view(-1, 1), local_rows)        # [n_recv, 1]
        emb = route.unpermute(route.to_requester(emb_owner))
        lin = route.unpermute(route.to_requester(lin_owner)).view(-1)
        ctx.route, ctx.local_rows = route, local_rows
        ctx.shapes = (tuple(W_local.shape), tuple(w1_local.shape))
        return emb, lin

    @staticmethod
    def backward(ctx, g_emb, g_lin):
        route, local_rows = ctx.route, ctx.local_rows
        Wshape, w1shape = ctx.shapes
        gW = gw1 = None
        # every rank's loss is a mean over ITS samples; like the all-reduce of the dense tail the
        # table gradients are averaged over ranks (gradient of the global-batch mean loss)
        inv = 1.0 / dist.get_world_size(route.group)
        if ctx.needs_input_grad[1]:
            g_rows = route.to_owner(g_emb.contiguous()[route.perm]).mul_(inv)
            gW = torch.sparse_coo_tensor(local_rows.view(1, -1), g_rows, Wshape, check_invariants=False)
        if ctx.needs_input_grad[2]:
            g1 = route.to_owner(g_lin.contiguous()[route.perm]).mul_(inv)
            gw1 = torch.sparse_coo_tensor(local_rows.view(1, -1), g1.view((-1,) + (1,) * (len(w1shape) - 1)),
                                          w1shape, check_invariants=False)
        return None, gW, gw1, None, None, None


def shard_rows(full: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """The rows of a full [N, ...] table that rank owns (row % world == rank), in local order."""
    return full[rank::world].contiguous()


def local_num_rows(N: int, rank: int, world: int) -> int:
    return (N - rank + world - 1) // world


class ShardedDeepFM(nn.Module):
    """DeepFM (src/models/deepfm.py:11-105) with row-sharded tables and a replicated MLP tail.

    Constructor arguments as the reference's DeepFM where they apply.  `embedding_shard`
    ([N_local, D]) and `fc_shard` ([N_local, 1]) hold only this rank's rows; use
    `load_full_tables` to shard a reference checkpoint's `embedding._emb_module.weight` /
    `fc.weight`.  Table gradients are row-form (sparse COO) on the local shards.
    """

    def __init__(self, field_dims: List[int], num_factor: int, hidden_sizes: List[int], p_dropout: float = 0.1,
                 use_batchnorm=False, device=None, process_group=None, gather: Optional[Callable] = None,
                 fm: Optional[Callable] = None, exact_routing: bool = False, bucket_slack: float = 1.25):
        """exact_routing=True sizes the all-to-all splits exactly (one host sync per step);
        the default pads every peer bucket to a fixed capacity (no sync, graph-capturable)."""
        super().__init__()
        self.exact_routing = exact_routing
        self.bucket_slack = bucket_slack
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.rank = dist.get_rank(process_group)
        self._gather = gather or _hip_gather
        self._fm = fm or self._hip_fm
        N = sum(field_dims)
        self.num_rows = N
        n_local = local_num_rows(N, self.rank, self.world)
        # xavier-uniform over the GLOBAL [N, D] matrix (src/models/embeddings/base.py:66-67)
        bound = (6.0 / (N + num_factor)) ** 0.5
        self.embedding_shard = nn.Parameter((torch.rand(n_local, num_factor, device=device) * 2 - 1) * bound)
        self.fc_shard = nn.Parameter(torch.randn(n_local, 1, device=device))   # N(0,1) like nn.EmbeddingBag
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
        self.register_buffer("bucket_overflow", torch.zeros(1, dtype=torch.bool, device=device), persistent=False)
        self.sync_dense_parameters()

    # ---- parameter plumbing ------------------------------------------------------------
    def dense_parameters(self):
        return [self._bias] + list(self._deep_branch.parameters())

    def sync_dense_parameters(self):
        """Replicas start from rank 0's dense weights."""
        for p in self.dense_parameters():
            dist.broadcast(p.data, src=dist.get_global_rank(self.group, 0) if self.group else 0, group=self.group)

    def allreduce_dense_grads(self):
        """Average the replicated tail's gradients with one flat collective."""
        ps = [p for p in self.dense_parameters() if p.grad is not None]
        if not ps:
            return
        flat = torch.cat([p.grad.reshape(-1) for p in ps])
        dist.all_reduce(flat, group=self.group)
        flat /= self.world
        o = 0
        for p in ps:
            n = p.numel()
            p.grad.copy_(flat[o:o + n].view_as(p.grad))
            o += n

    def check_overflow(self):
        """Synchronise; raise if any fixed-capacity bucket overflowed since the last check."""
        if bool(self.bucket_overflow.item()):
            self.bucket_overflow.zero_()
            raise RuntimeError("sharded lookup: a peer bucket overflowed its fixed capacity (extreme id skew); "
                               "raise bucket_slack or use exact_routing=True")

    @torch.no_grad()
    def load_full_tables(self, embedding_weight: torch.Tensor, fc_weight: torch.Tensor):
        self.embedding_shard.copy_(shard_rows(embedding_weight, self.rank, self.world))
        self.fc_shard.copy_(shard_rows(fc_weight, self.rank, self.world))

    # ---- forward -----------------------------------------------------------------------
    @staticmethod
    def _hip_fm(emb, lin, bias):
        # the first-order weights arrive already gathered: feed them to the fused FM kernel as
        # a B*F-row table addressed by the identity
        B, F, _ = emb.shape
        ident = torch.arange(B * F, device=emb.device).view(B, F)
        _, y = _kernels.fm_first_order(emb, ident, lin.reshape(-1, 1), bias)
        return y

    # ---- the compute between the collectives ------------------------------------------------
    def _local_compute(self, emb, lin):
        """emb [B,F,D], lin [B,F] (already exchanged) -> logits [B]: FM + first-order + MLP tail."""
        B = emb.shape[0]
        y_fm = self._fm(emb, lin, self._bias)
        if not emb.is_cuda:                                        # CPU only in the injected gloo tests
            return (y_fm.unsqueeze(1) + self._deep_branch(emb.reshape(B, -1))).squeeze(-1)
        return run_tail(self._deep_branch, emb.reshape(B, -1), last_add=y_fm).squeeze(-1)

    def enable_graphs(self, batch_size: int):
        """Capture the local compute (forward AND backward) for a fixed batch size as hipGraphs
        (torch.cuda.make_graphed_callables).  The RCCL collectives stay outside the graphs —
        recording them hung on this stack — so a step is: eager routing + all-to-alls, ONE graph
        replay for FM+MLP forward, one for their backward, eager gradient all-to-all + all-reduce."""
        F, D = self.offsets.shape[1], self.embedding_shard.shape[1]
        dev = self.embedding_shard.device
        # no collective may be in flight while a capture is open (the RCCL watchdog polls events)
        torch.cuda.synchronize(dev)
        dist.barrier(group=self.group)
        torch.cuda.synchronize(dev)
        emb = torch.randn(batch_size, F, D, device=dev, requires_grad=True)
        lin = torch.randn(batch_size, F, device=dev, requires_grad=True)

        class _Local(nn.Module):
            def __init__(inner, outer):
                super().__init__()
                inner.outer = [outer]            # not registered as a submodule (no param duplication)
                inner._bias = outer._bias
                inner._deep_branch = outer._deep_branch

            def forward(inner, e, l):
                return inner.outer[0]._local_compute(e, l)

        # kept out of the module registry (it shares this module's parameters; state_dict must not change)
        object.__setattr__(self, "_graphed_local",
                           torch.cuda.make_graphed_callables(_Local(self), (emb, lin), num_warmup_iters=3))
        object.__setattr__(self, "_graphed_batch", batch_size)
        torch.cuda.synchronize(dev)
        dist.barrier(group=self.group)

    def forward(self, x):
        """x: int [B_local, F] raw per-field ids -> logits [B_local]."""
        B, F = x.shape
        rows = x + self.offsets
        if self.exact_routing:
            emb, lin = ShardedLookup.apply(rows, self.embedding_shard, self.fc_shard, self.world, self.group,
                                           self._gather)
        else:
            emb, lin = ShardedLookupFixed.apply(rows, self.embedding_shard, self.fc_shard, self.world, self.group,
                                                self._gather, self.bucket_slack, self.bucket_overflow)
        emb, lin = emb.view(B, F, -1), lin.view(B, F)
        graphed = self.__dict__.get("_graphed_local")
        if graphed is not None and B == self._graphed_batch and self.training:
            return graphed(emb, lin)
        return self._local_compute(emb, lin)
