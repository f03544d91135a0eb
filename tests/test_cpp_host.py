"""The C++ host side (adaptiveviscositysolver_amd/host/avs_host.hpp + examples/hotpath_from_dump.cpp)
compiles and links against the C ABI; on a GPU it must reproduce the oracle through the dump format."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "adaptiveviscositysolver_amd")


def build_example(tmp_path):
    exe = str(tmp_path / "hotpath_from_dump")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           "-I", os.path.join(PKG, "host"), os.path.join(ROOT, "examples", "hotpath_from_dump.cpp"),
                           "-L", PKG, "-lavs_hip", f"-Wl,-rpath,{PKG}", "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    return exe


def test_cpp_host_compiles_and_links(tmp_path, built_lib):
    exe = build_example(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr


@pytest.mark.gpu
def test_cpp_host_reproduces_oracle(tmp_path, built_lib):
    from adaptiveviscositysolver_amd import scenes
    from adaptiveviscositysolver_amd.dump import write_dump
    from util import build_pyramid, feed, oracle_from_pyramid, rel_l2
    sc = scenes.sphere(32, 3)
    pyr = build_pyramid(sc)
    dump = str(tmp_path / "frame.avsd")
    write_dump(dump, sc, pyr)
    exe = build_example(tmp_path)
    out = str(tmp_path / "x.f64")
    r = subprocess.run([exe, dump, out, "1e-10", "5000"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    x = np.fromfile(out, dtype=np.float64)
    o = oracle_from_pyramid(sc, pyr)
    o.hot_path()
    xo, io = o.solve(1e-10, 5000)
    assert len(x) == len(xo) and rel_l2(x, xo) < 1e-5
    assert f"octree DOFS={len(xo)}" in r.stdout


@pytest.mark.gpu
def test_cpp_host_replays_a_non_power_of_two_frame(tmp_path, built_lib):
    """AVSDUMP2: a frame on a 112 x 48 x 48 simulation grid (the coarsened viscousBeam.hip equivalent; octree lattice 128 x 64 x 64)
    through the dump format and the C++ host -- round 2's writer refused every grid that is not a power of two."""
    import torch
    from adaptiveviscositysolver_amd import scenes
    from adaptiveviscositysolver_amd.dump import write_dump
    from util import build_pyramid, oracle_for_scene, rel_l2
    sc = scenes.viscous_beam_scene(coarsen=4)
    pyr = build_pyramid(scenes.to_device(sc, torch.device("cuda:0")))
    dump = str(tmp_path / "frame.avsd")
    write_dump(dump, sc, pyr)
    assert open(dump, "rb").read(8) == b"AVSDUMP2"
    exe = build_example(tmp_path)
    out = str(tmp_path / "x.f64")
    r = subprocess.run([exe, dump, out, "1e-10", "5000"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    x = np.fromfile(out, dtype=np.float64)
    o = oracle_for_scene(sc)           # the oracle's own pre-pass on the padded lattice
    o.prepass()
    o.hot_path()
    xo, io = o.solve(1e-10, 5000)
    assert len(x) == len(xo) and rel_l2(x, xo) < 1e-5
