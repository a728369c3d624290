"""bench.py's N > 1 plumbing on CPU (gloo, world_size 2): rank 0 generates the key set, every other
rank receives identical key material through broadcast_keys; the flat 65 536-gate batch is sharded over the
ranks by bench.shard (strong scaling) with no collective, so equality of keys and a
gap-free, overlap-free split are the whole contract."""
import hashlib
import os
import subprocess
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


def _bench(args, env_extra=None, drop=("RANK", "WORLD_SIZE", "LOCAL_RANK")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=300)


def test_gpus_flag_never_runs_on_fewer_gpus():
    """`bench.py --gpus N` is a complete N-GPU run or a loud failure: stand-alone it refuses when
    fewer than N devices are visible; under a launcher it refuses when WORLD_SIZE is not N.  Both exit with
    bench.EXIT_BAD_WORLD before anything is timed, and print no JSON line."""
    sys.path.insert(0, ROOT)
    import bench
    import torch

    visible = torch.cuda.device_count()
    r = _bench(["--gpus", str(visible + 2), "--steps", "1", "--cpu-sample", "0"])
    assert r.returncode == bench.EXIT_BAD_WORLD, (r.returncode, r.stderr[-400:])
    assert "GPU(s) visible" in r.stderr and "{" not in r.stdout
    r = _bench(["--gpus", "4", "--steps", "1", "--cpu-sample", "0"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"}, drop=())
    assert r.returncode == bench.EXIT_BAD_WORLD, (r.returncode, r.stderr[-400:])
    assert "WORLD_SIZE=2" in r.stderr and "{" not in r.stdout
    r = _bench(["--gpus", "0"])
    assert r.returncode == bench.EXIT_BAD_WORLD


def test_launcher_builds_a_torchrun_command(monkeypatch):
    """Stand-alone `--gpus 2` with two devices visible re-executes under torch.distributed.run with one rank per GPU
    and a loopback rendezvous; the child's exit status is passed on."""
    sys.path.insert(0, ROOT)
    import argparse
    import bench

    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    monkeypatch.setattr(bench.subprocess, "call", fake_call)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    args = argparse.Namespace(gpus=2, spawn=False)
    try:
        bench.launcher(args, script="/x/bench.py", argv=["--gpus", "2", "--steps", "5"], visible_gpus=2)
    except SystemExit as e:
        assert e.code == 7
    else:
        raise AssertionError("launcher returned instead of re-executing")
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=2" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-5:] == ["/x/bench.py", "--gpus", "2", "--steps", "5"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # one GPU, no --spawn: no re-execution, no process group
    assert bench.launcher(argparse.Namespace(gpus=1, spawn=False)) == (0, 1, 0, False)
    # under a launcher with the right world: this rank's coordinates
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "1")
    monkeypatch.setenv("LOCAL_RANK", "1")
    assert bench.launcher(args) == (1, 2, 1, True)


def test_counters_are_tied_to_the_build(tmp_path, monkeypatch):
    """bench.counters(): a committed counter file prices a live duration only when it was measured on the build that is
    loaded now: same workload + same build_id -> used; another build -> dropped, with the reason."""
    import argparse
    import json

    sys.path.insert(0, ROOT)
    import bench

    (tmp_path / "profiles").mkdir()
    rec = {"workload": {"gates_per_launch": 65536, "params": "128bit", "op": "NAND"}, "build_id": "aaaa", "valu_insts_per_launch": 1.0}
    (tmp_path / "profiles" / "c.json").write_text(json.dumps(rec))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "COUNTER_FILES", ("missing.json", "c.json"))
    args = argparse.Namespace(params="128bit", op="NAND")
    got, why = bench.counters(args, 65536, "aaaa", 3)
    assert got["valu_insts_per_launch"] == 1.0 and got["_file"] == "profiles/c.json" and why is None
    got, why = bench.counters(args, 65536, "bbbb", 3)
    assert got is None and "build aaaa" in why and "bbbb" in why
    got, why = bench.counters(args, 8192, "aaaa", 3)
    assert got is None and "no counter file" in why
    got, why = bench.counters(argparse.Namespace(params="80bit", op="NAND"), 65536, "aaaa", 4)
    assert got is None
    # counters of the 80-bit set's split decomposition (files older than the field: 4 levels) never price the direct one
    rec["workload"]["params"] = "80bit"
    (tmp_path / "profiles" / "c.json").write_text(json.dumps(rec))
    got, why = bench.counters(argparse.Namespace(params="80bit", op="NAND"), 65536, "aaaa", 4)
    assert got is not None
    got, why = bench.counters(argparse.Namespace(params="80bit", op="NAND"), 65536, "aaaa", 2)
    assert got is None
