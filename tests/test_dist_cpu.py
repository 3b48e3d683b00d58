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
            assert max(sizes) == (nb16 + n - 1) // n        # equal slab width everywhere
            sites = pdist.shard_sites(L, n)
            assert sum(hi - lo for lo, hi in sites) == L


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
