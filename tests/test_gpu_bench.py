"""bench.py as the driver starts it -- a plain subprocess, no torchrun, no environment -- on the GPU box: the launcher starts
the ranks itself, two ranks share the one GPU over the library's host transport (`--backend gloo`; RCCL refuses two ranks on
one device), and the line carries both scaling modes, the PARTIAL -> FINAL merge inside the step and Q3's exchange steps."""
import json

import pytest

from presto_amd import tpch
from tests.test_exchange_gloo import check_q1_q6, oracle_q1_q6, run_bench

pytestmark = pytest.mark.gpu


def test_bench_two_ranks_as_a_plain_subprocess(gpu, oracle):
    sf = 0.05
    r = run_bench(["--gpus", "2", "--backend", "gloo", "--sf", str(sf), "--steps", "2", "--warmup", "1", "--cpu-rows", "0"], timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    text = r.stdout.decode()
    assert text.count("\n") == 1, text[:2000]
    line = json.loads(text)
    rows = tpch.lineitem_rows(sf)
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["strong"]["scaling"] == "strong"
    assert line["config"]["rows_per_gpu"] == rows and line["strong"]["job_rows"] == rows
    assert abs(line["value"] - 2 * rows * 2 * 2 / (line["ms_per_step"] * 2 / 1e3)) <= 1e-6 * line["value"]
    check_q1_q6(line["results"], *oracle_q1_q6(oracle, sf * 2, rows * 2))
    check_q1_q6(line["strong"]["results"], *oracle_q1_q6(oracle, sf, rows))
    q3 = line["q3"]
    assert "error" not in q3 and q3["exchange"]["rank0_bytes_to_other_ranks_per_step"] > 0 and q3["exchange"]["transport"].startswith("host transport")
    assert q3["rank0"]["exchange_rows_sent"] > 0 and len(line["results"]["q3"]) == 10
    assert line["roofline"]["frac"] > 0 and line["roofline"]["launches"] > 0


def test_bench_one_rank_default_shape_small(gpu, oracle):
    """`--gpus 1`: no launcher, no merge; the line's side objects are all there."""
    sf = 0.05
    r = run_bench(["--gpus", "1", "--sf", str(sf), "--steps", "2", "--warmup", "1", "--cpu-rows", "200000", "--h2d-rows", "100000"], timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    line = json.loads(r.stdout.decode())
    rows = tpch.lineitem_rows(sf)
    assert line["n_gpus"] == 1 and "strong" not in line and line["config"]["rows_per_gpu"] == rows
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
