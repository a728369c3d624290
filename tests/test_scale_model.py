"""profiles/r05_scale_model.json is what tools/scale_model.py makes of profiles/r05_model_inputs.json (the 1-GPU measurements):
the committed table must be reproducible from the committed inputs, its 1-GPU column must agree with the committed bench lines,
and the flat configs must scale ~linearly (no exchange) while the netlists must not (the narrow levels' floor)."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _load(name):
    with open(os.path.join(ROOT, "profiles", name)) as f:
        return json.load(f)


def test_interp_is_piecewise_linear_and_extrapolates_proportionally():
    import scale_model

    pts = {"16": 1.0, "64": 2.0, "256": 8.0}
    assert scale_model.interp(pts, 8) == 1.0 and scale_model.interp(pts, 16) == 1.0
    assert scale_model.interp(pts, 40) == pytest.approx(1.5)
    assert scale_model.interp(pts, 160) == pytest.approx(5.0)
    assert scale_model.interp(pts, 512) == pytest.approx(16.0)


def test_committed_table_follows_from_the_committed_inputs():
    import scale_model

    inputs, table = _load("r05_model_inputs.json"), _load("r05_scale_model.json")
    a = table["assumptions"]
    again = scale_model.model(inputs, a["exchange_latency_us"], a["exchange_GBps"])
    for cfg, rec in table["configs"].items():
        for w, row in rec["by_gpus"].items():
            for k, v in row.items():
                assert again["configs"][cfg]["by_gpus"][w][k] == pytest.approx(v, rel=1e-9), (cfg, w, k)


def test_one_gpu_column_agrees_with_the_bench_lines_and_scaling_has_the_expected_shape():
    table = _load("r05_scale_model.json")["configs"]
    for cfg, bench in (("2_flat_nand_128bit", "r05_bench_final.json"), ("5_flat_nand_80bit", "r05_80bit_bench_final.json")):
        measured = _load(bench)["value"]
        rows = table[cfg]["by_gpus"]
        assert rows["1"]["strong_gates_per_s"] == pytest.approx(measured, rel=0.02)
        assert rows["8"]["strong_gates_per_s"] / rows["1"]["strong_gates_per_s"] > 7.8      # flat DAG: no exchange, whole rounds
    nets = {}
    with open(os.path.join(ROOT, "profiles", "r05_bench_netlist_balanced.txt")) as f:
        for line in f:
            d = json.loads(line)
            nets[d["net"]] = d["s_per_clock"]
    for cfg, net in (("3_mux_ram_8_16_16", "mux-ram"), ("4_cahp_system", "cahp-system")):
        rows = table[cfg]["by_gpus"]
        assert rows["1"]["s_per_clock"] == pytest.approx(nets[net], rel=0.04)
        speedup8 = rows["1"]["s_per_clock"] / rows["8"]["s_per_clock"]
        assert 2.0 < speedup8 < 4.0, (cfg, speedup8)      # deep, thin levels: one narrow-frontier pass per level on any number of GPUs
