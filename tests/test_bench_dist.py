"""bench.py's N > 1 plumbing on CPU (gloo, world_size 2): rank 0 generates the key set, every other
rank receives identical key material through broadcast_keys; the flat 65 536-gate batch is sharded over the
ranks by bench.shard (strong scaling) with no collective, so equality of keys and a
gap-free, overlap-free split are the whole contract."""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist

    sys.path.insert(0, ROOT)
    import bench
    from iyokan_amd import client
    from iyokan_amd.params import params_80bit

    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        p = params_80bit()
        keys = client.keygen(p, seed=1) if rank == 0 else bench.empty_keys(p)
        keys = bench.broadcast_keys(keys, dist, torch.device("cpu"), rank)
        h = hashlib.sha256()
        for name in ("s0", "s1", "bk", "ksk"):
            h.update(np.ascontiguousarray(getattr(keys, name)).tobytes())
        lo, cnt = bench.shard(65536, world, rank)
        t = torch.tensor([cnt], dtype=torch.int64)
        dist.all_reduce(t)                       # the shards of all ranks add up to the one batch
        q.put((rank, h.hexdigest(), int(keys.bk.any()), lo, cnt, int(t.item())))
    finally:
        dist.destroy_process_group()


def test_key_broadcast_two_ranks_gloo():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=10) for _ in range(2))
    assert got[0][1] == got[1][1] and got[1][2] == 1
    assert (got[0][3], got[0][4]) == (0, 32768) and (got[1][3], got[1][4]) == (32768, 32768)
    assert got[0][5] == got[1][5] == 65536


def test_shard_covers_the_batch_exactly():
    sys.path.insert(0, ROOT)
    import bench

    for total in (65536, 65537, 1000, 7):
        for world in (1, 2, 3, 4, 8):
            blocks = [bench.shard(total, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and sum(c for _, c in blocks) == total
            for (lo, c), (lo2, _) in zip(blocks, blocks[1:]):
                assert lo + c == lo2
            assert max(c for _, c in blocks) - min(c for _, c in blocks) <= 1
    assert bench.shard(65536, 8, 3) == (3 * 8192, 8192)
