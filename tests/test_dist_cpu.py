"""
N > 1 path on CPU: the shard partition and the exchange step (all-gather of per-shard slabs)
with torch.distributed / gloo, world_size 2 and 3.  The GPU-side use of the same exchange is
covered by tests/test_gpu_parity.py::test_sharded_evaluation_matches_single.
"""
import os
import socket

import numpy as np
import pytest

from evcouplings_amd import dist as pdist


def test_shard_partition_covers_all_sites_once():
    for L in (2, 15, 16, 17, 300, 500, 600, 1023):
        for n in (1, 2, 3, 4, 8):
            blocks = pdist.shard_blocks(L, n)
            assert len(blocks) == n
            nb16 = (L + 15) // 16
            covered = [b for lo, hi in blocks for b in range(lo, hi)]
            assert covered == list(range(nb16))
            sizes = [hi - lo for lo, hi in blocks]
            assert max(sizes) == (nb16 + n - 1) // n        # slab width
            assert max(sizes) - min(sizes) <= 1             # balanced: no idle shard unless n > nb16
            assert sizes == sorted(sizes)                   # the surplus blocks sit at the high end
            sites = pdist.shard_sites(L, n)
            assert sum(hi - lo for lo, hi in sites) == L
    assert [hi - lo for lo, hi in pdist.shard_blocks(300, 8)] == [2, 2, 2, 2, 2, 3, 3, 3]


def test_block_pair_ownership_is_a_balanced_partition():
    """Sharded-state mode, round 6: every block pair has exactly one owner (by construction of pair_owner), a shard owns
    the triangle over its own blocks + about half of every rectangle it shares; the busiest shard stays within one
    rectangle row per partner of the mean (round 5, pairs (I own, J >= I): 122 against 10 of 528 at L = 500 on 8 shards)."""
    for L, n in ((300, 8), (500, 8), (500, 2), (600, 4), (1000, 8), (48, 3), (20, 8)):
        parts = pdist.shard_blocks(L, n)
        nb16 = (L + 15) // 16
        counts = pdist.owned_block_pairs(L, n)
        assert sum(counts) == nb16 * (nb16 + 1) // 2
        cnt = [hi - lo for lo, hi in parts]
        for r in range(n):      # closed form used by the library (plm_half_blocks)
            mine = sum(((cnt[r] + 1) // 2) * cnt[p] if r < p else (cnt[p] // 2) * cnt[r] for p in range(n) if p != r)
            assert counts[r] == cnt[r] * (cnt[r] + 1) // 2 + mine
        mean = sum(counts) / n
        assert max(counts) <= mean + max(cnt) * (n - 1) / 2 + max(cnt) ** 2, (L, n, counts)
    assert pdist.owned_block_pairs(500, 8) == [66] * 8
    assert pdist.owned_block_pairs(500, 2) == [264, 264]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, per, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        buf = torch.zeros(world * per, dtype=torch.uint8)
        rng = np.random.default_rng(100 + rank)
        mine = torch.from_numpy(rng.integers(0, 256, size=per, dtype=np.uint8))
        buf[rank * per:(rank + 1) * per] = mine
        pdist.all_gather_inplace(buf, world, rank)
        q.put((rank, buf.numpy().copy()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_exchange_all_gather_gloo(world):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    per, port = 4096 + 256, _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, per, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = np.concatenate([np.random.default_rng(100 + r).integers(0, 256, size=per, dtype=np.uint8)
                             for r in range(world)])
    for r in range(world):
        np.testing.assert_array_equal(got[r], expect)     # every rank holds every shard's slab


def _coll_worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    from evcouplings_amd import _lib
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # all-to-all with uneven splits: rank r sends (r + 1) * (d + 2) * 8 bytes to rank d
        send_counts = [(rank + 1) * (d + 2) * 8 for d in range(world)]
        recv_counts = [(r + 1) * (rank + 2) * 8 for r in range(world)]
        send = torch.cat([torch.full((c,), 16 * rank + d, dtype=torch.uint8) for d, c in enumerate(send_counts)])
        recv = torch.zeros(sum(recv_counts), dtype=torch.uint8)
        pdist.collective_on_tensors(_lib.COLL_ALLTOALL, send, recv, send_counts, recv_counts)
        expect = torch.cat([torch.full((c,), 16 * r + rank, dtype=torch.uint8) for r, c in enumerate(recv_counts)])
        ok_a2a = bool((recv == expect).all())
        v64 = torch.arange(5, dtype=torch.float64) * (rank + 1)
        pdist.collective_on_tensors(_lib.COLL_ALLREDUCE_F64, v64.view(torch.uint8), None, [40], None)
        v32 = torch.ones(7, dtype=torch.float32) * (rank + 1)
        pdist.collective_on_tensors(_lib.COLL_ALLREDUCE_F32, v32.view(torch.uint8), None, [28], None)
        # broadcast of the parameter slices (final all-gather of J): every rank is root once
        ok_bc = True
        for root in range(world):
            b = torch.full((24,), 7 * root + 1 if rank == root else 0, dtype=torch.uint8)
            pdist.collective_on_tensors(_lib.COLL_BROADCAST, b, None, [24] * world, [root] * world)
            ok_bc = ok_bc and bool((b == 7 * root + 1).all())
        q.put((rank, ok_a2a and ok_bc, v64.tolist(), v32.tolist()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_state_collectives_gloo(world):
    """all_to_all_single with uneven byte splits and the f64 / f32 all-reduces of the sharded-state mode"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_coll_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    tot = world * (world + 1) / 2
    for rank, ok, v64, v32 in got:
        assert ok, "all-to-all payload wrong on rank %d" % rank
        assert v64 == [k * tot for k in range(5)] and v32 == [tot] * 7


def test_sharded_state_exchange_volumes_are_consistent():
    """what shard r sends to r' equals what r' expects from r (the byte counts the library derives)"""
    for L, n in ((300, 8), (500, 8), (40, 4), (100, 3)):
        blocks = pdist.shard_blocks(L, n)
        own = [hi - lo for lo, hi in blocks]
        for r in range(n):
            for rp in range(n):
                x_send = own[r] * own[rp] if rp > r else 0          # couplings go to higher shards
                x_recv_at_rp = own[rp] * own[r] if r < rp else 0
                assert x_send == x_recv_at_rp
        # every cross-shard block pair is exchanged exactly once in each direction
        nb = (L + 15) // 16
        cross = nb * (nb + 1) // 2 - sum(k * (k + 1) // 2 + k * sum(own[i + 1:]) - k * sum(own[i + 1:])
                                         for i, k in enumerate(own))
        assert cross == sum(own[r] * own[rp] for r in range(n) for rp in range(r + 1, n))


def _id_worker(rank, world, port, q):
    import torch.distributed as dist
    from evcouplings_amd import plm
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # the id itself needs a GPU (ncclGetUniqueId); what is tested here is that rank 0's bytes reach every rank
        plm.rccl_unique_id = lambda: bytes([17 + rank]) * plm.RCCL_ID_BYTES
        q.put((rank, pdist.share_rccl_id()))
    finally:
        dist.destroy_process_group()


def test_rccl_id_travels_from_rank_zero_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_id_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(3))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(got[r] == bytes([17]) * 128 for r in range(3))


def test_native_rccl_switch(monkeypatch):
    """the library-issued transport is opt-in (round 6, ADVICE r5: it has never run on more than one rank);
    PLM_NATIVE_RCCL=1 asks for it, and it is still only taken after a successful probe on every rank"""
    monkeypatch.delenv("PLM_NATIVE_RCCL", raising=False)
    assert not pdist.native_rccl_requested()
    monkeypatch.setenv("PLM_NATIVE_RCCL", "0")
    assert not pdist.native_rccl_requested()
    monkeypatch.setenv("PLM_NATIVE_RCCL", "1")
    assert pdist.native_rccl_requested()


def _negotiate_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        q.put((rank, pdist.negotiate_native_rccl()))
    finally:
        dist.destroy_process_group()


def test_negotiation_keeps_the_callback_transport_on_a_gloo_group():
    """negotiate_native_rccl is collective and must give every rank the same answer; a group whose backend is not nccl
    (the CPU / single-GPU flow tests) never takes the library-issued transport and never touches a GPU for the question"""
    import multiprocessing as mp
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_negotiate_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    assert got == {0: False, 1: False}
