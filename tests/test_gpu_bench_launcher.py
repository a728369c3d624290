"""GPU: the benches through their own launcher.  `--spawn` makes `--gpus 1` take the same path as
`--gpus 8`: re-execution under torch.distributed.run, an RCCL process group (one rank), key broadcast over it, barrier and
all-reduce / all-gather of the timings — the N > 1 code with N = 1, on the one GPU this box has."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, timeout):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    return subprocess.run([sys.executable] + cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_one_gpu_through_the_launcher():
    r = _run(["bench.py", "--gpus", "1", "--spawn", "--steps", "1", "--warmup", "0", "--gates", "4096", "--cpu-sample", "0"], 900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["rccl_world_size"] == 1 and len(line["per_rank_ms_per_step"]) == 1
    assert line["config"]["decrypt_check"] is True and line["value"] > 0
    r = line["roofline"]
    assert r["build_id"] and r["issue"]["useful_frac"] > 0
    # round 5: the headline is a compute fraction (issue slots with matching counters, else counted FP64 flops vs the 78.6 TFLOP/s
    # vector peak); the SURVEY 8(d) HBM figure stays beside it, marked non-binding
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and r["bound"] in ("valu-issue", "fp64-valu")
    assert r["hbm_contract"]["unit"] == "GB/s" and abs(r["contract_frac"] - r["hbm_contract"]["frac"]) < 1e-12
    assert r["fp64"]["unit"] == "TFLOP/s" and 0 < r["fp64"]["frac"] < 1 and r["fp64"]["peak"] == 78.6
    assert line["config"]["word_check"] is None   # 4096 gates: not the committed 65 536-gate workload


def test_bench_refuses_more_gpus_than_visible():
    import torch

    r = _run(["bench.py", "--gpus", str(torch.cuda.device_count() + 1), "--steps", "1", "--cpu-sample", "0"], 300)
    assert r.returncode == 3 and "GPU(s) visible" in r.stderr and "{" not in r.stdout


def test_netlist_bench_one_gpu_through_the_launcher():
    r = _run(["tools/bench_netlist.py", "--net", "counter", "--clocks", "1", "--gpus", "1", "--spawn"], 900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["rccl_world_size"] == 1 and line["outputs_match_plaintext"] is True
    # the per-level all_gather_into_tensor ran on RCCL, in place on the device arena, once per bootstrapped level
    assert line["collectives_per_clock"] > 0


def test_scale_all_writes_lines_and_stops_at_the_first_refusal(tmp_path):
    """tools/scale_all.sh (the one command for the 1/2/4/8 table): on this box it measures every GPU count that exists and
    exits with status 3 at the first count that does not, leaving the measured lines + a {"refused": ...} marker."""
    import torch

    n = torch.cuda.device_count()
    out = tmp_path / "scale.jsonl"
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update({"GPUS": f"{n} {n + 1}", "STEPS": "1"})
    r = subprocess.run(["bash", "tools/scale_all.sh", str(out)], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    lines = [json.loads(x) for x in out.read_text().splitlines()]
    assert r.returncode == 3, (r.returncode, r.stderr[-1000:])
    # line 0: the distinct-device tests (run where >= 2 GPUs are visible, a "skipped" record otherwise)
    assert "distinct_device_tests" in lines[0] and (n >= 2 or lines[0]["distinct_device_tests"].startswith("skipped"))
    assert n < 2 or lines[0]["status"] == 0, lines[0]
    assert len(lines) == 3 and lines[1]["n_gpus"] == n and lines[1]["config"]["decrypt_check"] is True
    assert lines[2]["refused"] == f"bench 128bit x{n + 1}" and lines[2]["status"] == 3
