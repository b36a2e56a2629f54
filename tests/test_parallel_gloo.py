"""not-gpu: the N > 1 host logic on two gloo ranks (CPU tensors): row sharding, the packed
[G | na | nb] all-reduce, and that the reduced Gram equals the single-process one."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, m, n0, n1, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from neuron_interpolation_model_fusion_amd import parallel
        g = torch.Generator().manual_seed(0)
        A = torch.randn(m, n0, generator=g, dtype=torch.float64)
        B = torch.randn(m, n1, generator=g, dtype=torch.float64)
        lo, hi = parallel.shard_rows(m)
        Al, Bl = A[lo:hi], B[lo:hi]
        G, na, nb = parallel.allreduce_gram(Al.T @ Bl, (Al * Al).sum(0), (Bl * Bl).sum(0))
        ok = torch.allclose(G, A.T @ B, rtol=1e-12) and torch.allclose(na, (A * A).sum(0), rtol=1e-12) \
            and torch.allclose(nb, (B * B).sum(0), rtol=1e-12)
        # the context manager routes the same reduction, and is a no-op outside
        with parallel.SampleSharding():
            G2, _, _ = parallel.SampleSharding.reduce(Al.T @ Bl)
        ok = ok and torch.allclose(G2, A.T @ B, rtol=1e-12)
        G3, _, _ = parallel.SampleSharding.reduce(Al.T @ Bl)
        ok = ok and torch.equal(G3, Al.T @ Bl)
        # every rank ends with bit-identical buffers (replicated assignment needs no broadcast)
        gathered = [torch.empty_like(G) for _ in range(world)]
        dist.all_gather(gathered, G.contiguous())
        ok = ok and all(torch.equal(gathered[0], t) for t in gathered)
        out[rank] = bool(ok) and (lo, hi) == (m * rank // world, m * (rank + 1) // world)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('m,n0,n1', [(101, 7, 5), (64, 16, 16)])
def test_sharded_gram_allreduce_two_ranks(m, n0, n1):
    world = 2
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), m, n0, n1, out), nprocs=world, join=True)
        assert dict(out) == {0: True, 1: True}


def test_shard_rows_cover_everything():
    from neuron_interpolation_model_fusion_amd.parallel import shard_rows, pack_gram, unpack_gram
    for m in (1, 7, 325000):
        for world in (1, 2, 3, 8):
            spans = [shard_rows(m, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == m
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    G, na, nb = torch.arange(6.).view(2, 3), torch.tensor([1., 2.]), torch.tensor([3., 4., 5.])
    g2, a2, b2 = unpack_gram(pack_gram(G, na, nb), 2, 3)
    assert torch.equal(g2, G.double()) and torch.equal(a2, na.double()) and torch.equal(b2, nb.double())


def _worker_sym(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from neuron_interpolation_model_fusion_amd import parallel
        parallel.TRI_BLOCK = 64  # small blocks so that a small matrix takes the packed path
        g = torch.Generator().manual_seed(1)
        m, n = 90, 203  # ragged last block
        A = torch.randn(m, n, generator=g, dtype=torch.float64)
        lo, hi = parallel.shard_rows(m)
        Al = A[lo:hi]
        Gl, nal = Al.T @ Al, (Al * Al).sum(0)
        Gl = torch.triu(Gl) + torch.triu(Gl, 1).T  # exactly symmetric partial, as the Gram kernel produces
        full, na_full, _ = parallel.allreduce_gram(Gl.clone(), nal.clone(), nal.clone())
        sym, na_sym, nb_sym = parallel.allreduce_gram(Gl.clone(), nal.clone(), nal.clone(), symmetric=True)
        ok = torch.equal(sym, full) and torch.equal(na_sym, na_full) and torch.equal(nb_sym, na_full)
        ok = ok and torch.equal(sym, sym.T)
        # round trip of the packing alone
        buf = parallel.pack_gram_upper(Gl, nal)
        assert buf.numel() < n * n + n
        G2, na2 = parallel.unpack_gram_upper(buf, n)
        ok = ok and torch.equal(G2, Gl) and torch.equal(na2, nal)
        out[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_symmetric_gram_reduces_only_the_block_upper_triangle():
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker_sym, args=(2, _free_port(), out), nprocs=2, join=True)
        assert dict(out) == {0: True, 1: True}


def _worker_gather(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from neuron_interpolation_model_fusion_amd import parallel
        full = torch.arange(7 * 3 * 2, dtype=torch.float32).view(7, 3, 2)   # 7 rows over 2 ranks: shards of 3 and 4
        lo, hi = parallel.shard_rows(7)
        got = parallel.allgather_rows(full[lo:hi].clone())
        ok = torch.equal(got, full)
        lin = torch.nn.Linear(3, 2)
        with torch.no_grad():
            lin.weight.fill_(float(rank + 1))
        parallel.broadcast_parameters(lin, 0)
        ok = ok and bool((lin.weight == 1.0).all())
        out[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


def test_allgather_rows_restores_ragged_row_shards_and_broadcast_aligns_parameters():
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker_gather, args=(2, _free_port(), out), nprocs=2, join=True)
        assert dict(out) == {0: True, 1: True}
    # outside a process group both are the identity
    from neuron_interpolation_model_fusion_amd import parallel
    t = torch.randn(5, 2)
    assert parallel.allgather_rows(t) is t
