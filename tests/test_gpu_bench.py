"""bench.py as the driver starts it -- a plain subprocess, no torchrun, no environment -- on the GPU box: the launcher starts
the ranks itself, two ranks share the one GPU over the library's host transport (`--backend gloo`; RCCL refuses two ranks on
one device), and the line carries both scaling modes, the PARTIAL -> FINAL merge inside the step and Q3's exchange steps."""
import json

import pytest

from presto_amd import tpch
from tests.test_exchange_gloo import check_line_shape, check_q1_q6, oracle_q1_q6, run_bench

pytestmark = pytest.mark.gpu


def test_bench_two_ranks_as_a_plain_subprocess(gpu, oracle, tmp_path):
    """Two ranks on the one GPU over the library's host transport.  The ranks never import torch (the detail file says so, and a rank
    that finds it imported leaves with exit code 5): control plane = presto_amd/control.py, one ROCm stack per process."""
    sf = 0.05
    path = str(tmp_path / "detail.json")
    r = run_bench(["--gpus", "2", "--backend", "gloo", "--sf", str(sf), "--steps", "2", "--warmup", "1", "--cpu-rows", "0", "--detail", path], timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    line = check_line_shape(r.stdout.decode())
    d = json.load(open(path))
    rows = tpch.lineitem_rows(sf)
    assert d["torch_imported"] is False and d["config"]["control_plane"].startswith("local socket")
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and d["weak"]["scaling"] == "weak" and line["weak"]["value"] > 0
    assert d["weak"]["rows_per_gpu_rank0"] == rows and d["config"]["rows_per_gpu"] == rows // 2 // 4 * 4
    assert abs(d["value"] - 2 * rows * 2 / (d["ms_per_step"] * 2 / 1e3)) <= 1e-6 * d["value"]
    check_q1_q6(d["results"], *oracle_q1_q6(oracle, sf, rows))
    check_q1_q6(d["weak"]["results"], *oracle_q1_q6(oracle, sf * 2, rows * 2))
    q3 = d["q3"]
    assert "error" not in q3 and q3["exchange"]["rank0_bytes_to_other_ranks_per_step"] > 0 and q3["exchange"]["transport"].startswith("host transport")
    assert q3["rank0"]["exchange_rows_sent"] > 0 and len(d["results"]["q3"]) == 10
    assert line["roofline"]["frac"] > 0 and line["roofline"]["launches"] > 0 and line["roofline"]["kernel"].startswith("pa_fused_lds_")


def test_bench_goes_on_over_the_host_transport_when_rccl_refuses(gpu, oracle, tmp_path):
    """`--backend nccl` with two ranks on the one GPU: RCCL refuses the communicator (two ranks on one device) with an error on every
    rank.  The ranks agree over the control plane, go on over the host transport and still deliver the headline -- both scaling modes,
    results equal to the oracle's -- without the Q3 leg; the line says what happened (config.data_plane, q3.error) and the exit code is 0."""
    sf = 0.05
    path = str(tmp_path / "detail.json")
    r = run_bench(["--gpus", "2", "--backend", "nccl", "--sf", str(sf), "--steps", "2", "--warmup", "1", "--cpu-rows", "0", "--preflight-timeout", "60",
                   "--detail", path], timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    line = check_line_shape(r.stdout.decode())
    d = json.load(open(path))
    rows = tpch.lineitem_rows(sf)
    assert d["torch_imported"] is False
    assert d["config"]["data_plane"].startswith("host transport over the control plane -- RCCL failed"), d["config"]["data_plane"]
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["value"] > 0 and d["weak"]["value"] > 0
    assert "error" in d["q3"] and "error" in line["q3"]
    check_q1_q6(d["results"], *oracle_q1_q6(oracle, sf, rows))
    check_q1_q6(d["weak"]["results"], *oracle_q1_q6(oracle, sf * 2, rows * 2))


def test_bench_one_rank_default_shape_small(gpu, oracle, tmp_path):
    """`--gpus 1`: no launcher, no merge; the side legs are all in the detail file, their one-number summaries in the line."""
    sf = 0.05
    path = str(tmp_path / "detail.json")
    r = run_bench(["--gpus", "1", "--sf", str(sf), "--steps", "2", "--warmup", "1", "--cpu-rows", "200000", "--h2d-rows", "100000", "--detail", path],
                  timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    summary = check_line_shape(r.stdout.decode())
    line = json.load(open(path))
    assert line["line"] == summary
    rows = tpch.lineitem_rows(sf)
    assert line["n_gpus"] == 1 and "strong" not in line and "weak" not in line and line["config"]["rows_per_gpu"] == rows
    assert summary["roofline"]["kernel"].startswith("pa_fused_lds_") and summary["roofline_q6"]["kernel"].startswith("pa_fused_global_")
    assert summary["cpu_baseline"]["value"] > 0 and summary["cpu_baseline"]["cores"] >= 1 and summary["q3"]["cpu_rows_s"] > 0
    assert len(summary["operators"]) == 5 + 16 + 2 and all(v[0] > 0 and 0 < v[1] < 1 for v in summary["operators"].values())
    check_q1_q6(line["results"], *oracle_q1_q6(oracle, sf, rows))
    assert "error" not in line["q3"] and line["q3"]["exchange"] == "none (one rank)"
    assert line["cpu_baseline"]["value"] > 0 and line["h2d"]["value"] > 0
    assert line["cpu_baseline"]["cores"] == line["cpu_baseline"]["physical_cores"] >= 1 and line["cpu_baseline"]["runs"] >= 10
    assert line["q3"]["cpu_baseline"]["value"] > 0
    ops = line["operators"]
    assert "error" not in ops, ops
    assert [e["groups"] for e in ops["hash_agg"]] == [3_000_000, 4, 1000, 100_000, 3_000_000] and ops["hash_agg"][0]["rows"] == 10_000_000
    for e in ops["hash_agg"]:
        assert e["groups_out"] <= e["groups"] and e["value"] > 0 and 0 < e["frac"] < 1 and e["cpu"]["value"] > 0
    assert len(ops["hash_join"]) == 8
    for e in ops["hash_join"]:
        assert e["build"]["value"] > 0 and e["probe"]["value"] > 0 and e["probe"]["cpu"]["value"] > 0
    # the reference's probe pages: every row matches at rate 1 (x1 build rows), about a tenth at 0.1, five build rows per match at x5
    by_case = {e["case"]: e for e in ops["hash_join"]}
    assert by_case["8 M build rows x1, 1.4 M probe rows, match rate 1"]["probe"]["matches"] == 1_400_000
    assert 120_000 < by_case["8 M build rows x1, 1.4 M probe rows, match rate 0.1"]["probe"]["matches"] < 160_000
    assert 6_900_000 < by_case["8 M build rows x5, 1.4 M probe rows, match rate 1"]["probe"]["matches"] < 7_100_000
    assert ops["order_by"]["value"] > 0 and ops["topn"]["value"] > 0 and ops["topn"]["cpu"]["value"] > 0
