"""The JSON line bench.py prints (the driver's contract): checked on the committed line of the final build, so that a
change of bench.py that drops or renames a field shows up without a GPU."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _latest_default_line():
    tag = open(os.path.join(ROOT, "profiles", "FINAL")).read().strip()   # tag of the final build of the round
    return json.load(open(os.path.join(ROOT, "profiles", f"r01_{tag}_bench.json")))


def test_bench_line_has_the_contract_fields():
    r = _latest_default_line()
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in r, key
    assert r["metric"] == "lattice-site-updates/sec" and r["higher_is_better"] is True and r["scaling"] == "weak"
    assert r["vs_baseline"] is None and r["dtype"] == "f64" and r["data"] == "synthetic"
    assert "workload" in r["config"] and "model" not in r["config"]
    roof = r["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in roof, key
    assert roof["bound"] in ("hbm", "mfma") and roof["unit"] in ("GB/s", "TFLOP/s")
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    cpu = r["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cpu, key
    assert cpu["kind"] in ("reference", "port")
    # value = units of all ranks / time
    assert abs(r["value"] * r["ms_per_step"] * 1e-3 / (2 * 1024 * 1024 * 11 * r["config"]["chains_total"]) - 1.0) < 1e-6


def test_bench_source_keeps_the_oracle_out_of_the_timed_path():
    src = open(os.path.join(ROOT, "bench.py")).read()
    # the oracle is only reached through the cpu_baseline subprocess
    assert "import oracle" not in src and "from oracle" not in src
    assert "cpu_baseline.py" in src
