"""bench.py must be launchable exactly as the driver types it: `python bench.py --gpus N ...` with no launcher
environment starts its own N ranks (torch.distributed.run on 127.0.0.1).  Checked on the CPU with a 2-rank gloo run that
goes up to -- not including -- avs_dist_init (--launch-check).  Also: the clean-subprocess CPU baseline runner."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    return env


def test_launcher_command_is_the_drivers_form():
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.launcher_command(8, ["--gpus", "8", "--steps", "2"], port=29512)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert "--nproc-per-node=8" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29512"
    assert cmd[-4:] == ["--gpus", "8", "--steps", "2"] and cmd[-5].endswith("bench.py")


def test_plain_python_invocation_spawns_its_ranks():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], env=clean_env(),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                     # rank 0 prints exactly one JSON line
    out = json.loads(lines[0])
    assert out == {"launch_check": True, "n_gpus": 2, "backend": "gloo"} or (out["launch_check"] and out["n_gpus"] == 2)


def test_under_a_launcher_it_joins_instead_of_spawning():
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.launcher_command(2, ["--gpus", "2", "--launch-check"])
    r = subprocess.run(cmd, env=clean_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert sum(1 for l in r.stdout.splitlines() if l.startswith("{")) == 1


def test_cpu_baseline_runner_reports_both_variants(tmp_path):
    """oracle/cpu_baseline.py in a clean subprocess: eigen_faithful (serial vector ops) and all_parallel"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from adaptiveviscositysolver_amd import scenes
    from util import oracle_for_scene
    o = oracle_for_scene(scenes.fat_beam(32, 2))
    o.prepass()
    o.hot_path()
    A = o.csr()
    for name, arr in (("row_ptr", A.row_ptr), ("col", A.col), ("val", A.val), ("rhs", A.rhs), ("x0", o.initial_guess())):
        np.save(tmp_path / (name + ".npy"), arr)
    env = dict(clean_env(), OMP_NUM_THREADS="2", OMP_PROC_BIND="close", OMP_PLACES="cores")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline.py"), str(tmp_path), "1e-3", "2", "2"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    ef, ap = out["variants"]["eigen_faithful"], out["variants"]["all_parallel"]
    assert ef["vector_threads"] == 1 and ap["vector_threads"] == 2 and ef["spmv_threads"] == 2
    assert ef["iterations"] == ap["iterations"] > 0 and ef["spmv_gbps"] > 0
    assert out["cpu_model"] and out["n"] == len(A.rhs)
    assert "torch" not in r.stderr
