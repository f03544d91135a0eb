"""bench.py's stdout line must stay parseable by the driver: <= 4 KB, strict JSON (round-5 review: a 28 KB line was lost).

The reference's own observable is one short line (iterations, error, DOFs: HDK_AdaptiveViscosity.cpp:645-652)."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _raise(name):
    raise ValueError("non-strict JSON constant " + name)


def synthetic_full(nan=False, long=4000, extras=12):
    blob = "x" * long
    roof = {"bound": "hbm", "kernel": "k_spmv_brick<DOT> " + blob, "achieved": 2675.6, "peak": 8000.0, "unit": "GB/s", "frac": 0.334,
            "traffic": float("nan") if nan else 2.95e8, "traffic_source": blob, "stored_bytes_per_launch": 241234680,
            "algorithmic_bytes_per_launch": 1491589956, "effective_gbps": float("inf") if nan else 13507.6, "mean_launch_us": 110.4,
            "binding": {"resource": "valu_issue " + blob, "frac": 0.58, "source": blob}}
    return {"metric": "cg_iterations_per_sec", "value": 5060.9, "unit": "iter/s", "n_gpus": 1, "steps": 20, "warmup": 5,
            "ms_per_step": 251.3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "fat_beam 512^3 " + blob, "baseline_config": 4, "n_dofs": 7419676, "nnz": 111933036,
                       "cg_iterations_per_step": 1272, "parallelism": "single"},
            "roofline": roof,
            "cpu_baseline": {"value": 25.97, "unit": "iter/s", "cores": 16, "threads": 16, "kind": "port", "variant": blob,
                             "cpu_model": "AMD EPYC 9575F 64-Core Processor", "sample": blob, "all_parallel": {"iter_per_s": 53.7},
                             "eigen_faithful": {"iter_per_s": 25.97}},
            "assembly_ms": {"wall": 13.6}, "hot_path_ms": 264.9, "speedup_vs_cpu_baseline": 194.8,
            "dist": {"transport": "direct", "verification": [{"ok": True, "solves": [blob]}]},
            "extra_workloads": [{"workload": f"workload {i} " + blob, "dtype": "f64", "value": 1000.0 * i, "ms_per_step": 10.0,
                                 "roofline": dict(roof)} for i in range(extras)]}


@pytest.mark.parametrize("nan", [False, True])
@pytest.mark.parametrize("extras", [0, 8, 40])
def test_headline_is_short_strict_json(nan, extras):
    h = bench.headline_of(synthetic_full(nan=nan, extras=extras), "bench_extra.json")
    line = json.dumps(h, allow_nan=False)
    assert len(line) + 1 < 4096
    back = json.loads(line, parse_constant=_raise)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in back
    assert back["config"]["workload"].startswith("fat_beam 512^3")
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in back["roofline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in back["cpu_baseline"]
    assert len(back["roofline"]["kernel"]) <= 80


def test_emit_writes_the_full_record_and_one_short_line(tmp_path, monkeypatch, capsys):
    monkeypatch.setattr(bench, "FULL_RECORD", str(tmp_path / "bench_extra.json"))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    bench.emit(synthetic_full(nan=True))
    out = capsys.readouterr().out
    assert out.count("\n") == 1 and len(out) < 4096
    json.loads(out, parse_constant=_raise)
    full = json.loads(open(tmp_path / "bench_extra.json").read(), parse_constant=_raise)
    assert len(full["extra_workloads"]) == 12 and "notes" in full


def test_the_last_committed_round_record_digests_under_the_limit():
    p = os.path.join(ROOT, "profiles", "r05b_bench_final.json")
    full = json.load(open(p))
    line = json.dumps(bench.headline_of(full, "bench_extra.json"), allow_nan=False)
    assert len(line) < 4096
