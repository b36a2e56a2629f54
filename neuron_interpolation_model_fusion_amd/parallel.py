"""Sample-sharded fusion over one 8 x MI355X node (no counterpart in the reference, which is
single-process: SURVEY section 5 / 8e).

The fusion samples (rows of every activation matrix) are split over the ranks, one process per
GPU.  Every quantity of the path that depends on the samples is a sum over rows -- A^T B, X^T X,
X^T A, squared column norms -- so each rank builds its partial with the local kernels and ONE
all-reduce (sum, f64) per level combines the packed [G | na | nb] buffer.  f64 keeps the
rank-order of the sum from touching the 1e-6 tie margin of the assignments.  What follows the
reduction (cost epilogue, assignment, k-means on the N x N Gram, refit) is small and replicated:
every rank computes the identical result from the identical reduced buffer, so no broadcast is
needed.  torch.distributed's "nccl" backend is RCCL on ROCm; "gloo" runs the same code on CPU
tensors (tests/test_parallel_gloo.py).
"""
import os

import torch
import torch.distributed as dist


def shard_rows(n_rows, rank=None, world=None):
    """Half-open row range [lo, hi) of this rank: contiguous, balanced to within one row."""
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    return n_rows * rank // world, n_rows * (rank + 1) // world


def pack_gram(G, na=None, nb=None):
    """One contiguous f64 buffer [G.ravel() | na | nb] (a single collective per level)."""
    parts = [G.reshape(-1).double()]
    if na is not None:
        parts.append(na.double())
    if nb is not None:
        parts.append(nb.double())
    return torch.cat(parts)


def unpack_gram(buf, n0, n1, has_na=True, has_nb=True):
    G = buf[:n0 * n1].view(n0, n1)
    off = n0 * n1
    na = nb = None
    if has_na:
        na = buf[off:off + n0]
        off += n0
    if has_nb:
        nb = buf[off:off + n1]
    return G, na, nb


TRI_BLOCK = 1024  # block rows of the packed upper triangle


def _via_host(t, group=None):
    """gloo only (the CPU-side rehearsal backend; RCCL takes device tensors and orders itself on the stream): device
    tensors go through the collective as host copies.  gloo's own device path stages through the host on streams of
    its own, and two ranks that share one GPU (tests/test_gpu_sharded.py, the bench rehearsal) were seen to stall
    inside it for minutes, waiting on each other's queued kernels (profiles/r2_gloo2_stall_stacks_before_fix.txt).
    The .cpu() copy also finishes this rank's queued work on the tensor first."""
    return t.is_cuda and dist.get_backend(group) != 'nccl'


DIRECT_MIN_BYTES = 8 << 20  # below this one library all-reduce is latency-bound either way


def _direct_all_reduce(buf, group=None):
    """Sum over the group as a DIRECT reduce-scatter + all-gather: the 8 GPUs of a node are a full
    xGMI mesh, 7 links x ~153 GB/s each, so every rank sends chunk r of its buffer straight to rank r (all links busy
    at once, one hop), sums the chunks it received in rank order, and sends its reduced chunk straight to everybody --
    2 (S / W) bytes per link against the 2 (W - 1) (S / W) of a ring, which is per-link bound on a mesh.  The sum is
    formed in one fixed order on one rank per chunk, so every rank ends with the same bits whatever the library would
    have chosen.  Two all-to-alls (RCCL: concurrent point-to-point sends)."""
    W = dist.get_world_size(group)
    n = buf.numel()
    chunk = (n + W - 1) // W
    send = buf if n == W * chunk else torch.cat([buf, buf.new_zeros(W * chunk - n)])
    recv = torch.empty_like(send)
    dist.all_to_all_single(recv, send, group=group)            # recv[r] = chunk `me` of rank r
    mine = recv.view(W, chunk)[0].clone()
    for r in range(1, W):                                       # rank order: a fixed summation order
        mine += recv.view(W, chunk)[r]
    out = torch.empty_like(send)
    dist.all_to_all_single(out, mine.repeat(W), group=group)   # out[r] = reduced chunk r, straight from its owner
    buf.copy_(out[:n])
    return buf


def all_reduce_sum(buf, group=None):
    """In-place sum of `buf` over the group.  RCCL: large buffers go through the direct reduce-scatter + all-gather
    when NIF_ALLREDUCE=direct (off by default: it could only be rehearsed with one rank on the one-GPU box; the
    library all-reduce is the path every multi-rank test here exercises)."""
    if _via_host(buf, group):
        host = buf.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        buf.copy_(host)
    elif (os.environ.get('NIF_ALLREDUCE') == 'direct' and buf.is_cuda and
          buf.numel() * buf.element_size() >= DIRECT_MIN_BYTES):
        _direct_all_reduce(buf, group)
    else:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    return buf


def _all_gather(mine, group=None):
    world = dist.get_world_size(group)
    if _via_host(mine, group):
        host = mine.cpu()
        parts = [torch.empty_like(host) for _ in range(world)]
        dist.all_gather(parts, host, group=group)
        return [p.to(mine.device) for p in parts]
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)
    return parts


def pack_gram_upper(G, na=None, block=None):
    """Symmetric G: only the block upper triangle travels (rows [i, i + block) from column i on), about half
    of the N^2 entries for N >> block -- at N = 24 576 (C5) 2.5 GB of f64 instead of 4.8 GB."""
    n = G.shape[0]
    block = block or TRI_BLOCK
    parts = [G[i:i + block, i:].reshape(-1).double() for i in range(0, n, block)]
    if na is not None:
        parts.append(na.double())
    return torch.cat(parts)


def unpack_gram_upper(buf, n, has_na=True, block=None):
    """Inverse of pack_gram_upper; the strictly lower blocks are the transposes of the reduced upper ones, so the
    result is exactly symmetric and equal, entry by entry, to reducing the full matrix."""
    block = block or TRI_BLOCK
    G = torch.empty((n, n), dtype=buf.dtype, device=buf.device)
    off = 0
    for i in range(0, n, block):
        rows = min(block, n - i)
        blk = buf[off:off + rows * (n - i)].view(rows, n - i)
        G[i:i + rows, i:] = blk
        if i + rows < n:
            G[i + rows:, i:i + rows] = blk[:, rows:].T
        off += rows * (n - i)
    na = buf[off:off + n] if has_na else None
    return G, na


def allreduce_gram(G, na=None, nb=None, group=None, symmetric=False):
    """Sum the per-rank partial Gram quantities over the group; returns (G, na, nb) of the reduced
    buffer.  A no-op when torch.distributed is not initialised or the group has one rank.
    symmetric=True (G = A^T A): the packed block upper triangle is reduced and mirrored."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return G, na, nb
    n0, n1 = G.shape
    if symmetric and n0 == n1 and n0 > 2 * TRI_BLOCK:
        buf = all_reduce_sum(pack_gram_upper(G, na), group)
        G, na = unpack_gram_upper(buf, n0, na is not None)
        return G, na, (na if nb is not None else None)
    buf = all_reduce_sum(pack_gram(G, na, nb), group)
    return unpack_gram(buf, n0, n1, na is not None, nb is not None)


def allgather_rows(t, group=None):
    """Concatenation over the ranks (in rank order) of every rank's rows of `t` -- the row shards of one
    sample-indexed tensor put back together on every rank.  Shards may differ in length (shard_rows balances
    them to within one row): they are padded to the longest for the collective.  A no-op outside a group."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return t
    rows = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device if dist.get_backend(group) == 'nccl' else 'cpu')
    counts = [int(c.item()) for c in _all_gather(rows, group)]
    longest = max(counts)
    mine = t.contiguous()
    if mine.shape[0] < longest:
        pad = torch.zeros((longest - mine.shape[0],) + tuple(mine.shape[1:]), dtype=mine.dtype, device=mine.device)
        mine = torch.cat([mine, pad], dim=0)
    parts = _all_gather(mine, group)
    return torch.cat([p[:c] for p, c in zip(parts, counts)], dim=0)


def broadcast_parameters(module, src=0, group=None):
    """Overwrite every parameter and buffer of `module` with rank `src`'s (after a replicated computation that
    must not drift between ranks)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    with torch.no_grad():
        root = dist.get_global_rank(group, src) if group is not None else src
        for t in list(module.parameters()) + list(module.buffers()):
            if _via_host(t, group):
                host = t.data.cpu()
                dist.broadcast(host, src=root, group=group)
                t.data.copy_(host)
            else:
                dist.broadcast(t.data, src=root, group=group)


class SampleSharding:
    """Context: inside it, ops.xty calls whose contraction axis is the sample axis reduce their
    outputs over `group` (the host classes mark those calls with rows_sharded=True).

        with SampleSharding(group):          # every rank holds its shard of X
            fused = GitRebasin(m0, m1).fuse(X_local, None, 'uniform')
    """
    _active = None

    def __init__(self, group=None, shard_lloyd=None):
        self.group = group
        # opt-in: the k-means Lloyd steps of HF / KF / ALF split their pass over the (reduced, replicated) Gram by
        # neuron as well, one small all-reduce per step (ops.KMeans.lloyd_step); same bits as the replicated step
        self.shard_lloyd = os.environ.get('NIF_SHARD_LLOYD') == '1' if shard_lloyd is None else bool(shard_lloyd)

    def __enter__(self):
        self._prev = SampleSharding._active
        SampleSharding._active = self
        return self

    def __exit__(self, *exc):
        SampleSharding._active = self._prev
        return False

    @classmethod
    def lloyd_blocks(cls, nblocks):
        """(lo, hi, group): this rank's contiguous share of `nblocks` 64-neuron blocks in a neuron-sharded Lloyd step,
        or None when the step is to run whole (no context, not opted in, one rank)."""
        act = cls._active
        if act is None or not act.shard_lloyd or not dist.is_initialized():
            return None
        world = dist.get_world_size(act.group)
        if world <= 1:
            return None
        per = -(-nblocks // world)
        r = dist.get_rank(act.group)
        return min(nblocks, r * per), min(nblocks, (r + 1) * per), act.group

    @classmethod
    def wants_packed(cls, n):
        """A symmetric n x n Gram of a sharded run travels as a packed block upper triangle (written by the reduction
        kernel, ops.xty_sym_packed) when it is large enough for the halved traffic to matter."""
        return (cls._active is not None and dist.is_initialized() and dist.get_world_size(cls._active.group) > 1 and
                n > 2 * TRI_BLOCK)

    @classmethod
    def reduce(cls, G, na=None, nb=None, symmetric=False):
        if cls._active is None:
            return G, na, nb
        return allreduce_gram(G, na, nb, cls._active.group, symmetric=symmetric)
