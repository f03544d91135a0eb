"""Post-solve transfer on the device (avs_transfer_to_regular_grid) against the oracle: regular-grid
classification, interpolator node grids and the final regular MAC-grid velocity.  Node values and the
output are fp32 fields fed by double arithmetic in the reference's order: required bit-exact when both
sides start from the same solution vector."""
import ctypes as C

import numpy as np
import pytest
import torch

from adaptiveviscositysolver_amd import DevicePrepass, ViscositySolve, capi, scenes
from oracle import oracle as O
from util import oracle_for_scene, rel_l2

pytestmark = pytest.mark.gpu

CASES = {
    "sphere32_L3": lambda dev: scenes.sphere(32, 3, device=dev),
    "beam64_L4": lambda dev: scenes.fat_beam(64, 4, device=dev),
    "beam64_wall_varvisc": lambda dev: scenes.fat_beam(64, 3, wall=True, variable_viscosity=True, device=dev),
    "sheet64": lambda dev: scenes.thin_sheet(64, 3, thickness_cells=12, device=dev),
    "noncubic": lambda dev: scenes.fat_beam(64, 3, res=(64, 32, 32), device=dev),
}


def gpu_pipeline(sc):
    pp = DevicePrepass(sc.res, sc.dx, sc.levels)
    info = pp.run(sc.liquid, sc.solid)
    s = ViscositySolve(sc.res, sc.dx, sc.dt, info.levels, device=0)
    pp.apply(s)
    s.set_scene_fields(sc)
    s.assemble()
    return pp, info, s


@pytest.mark.parametrize("name", list(CASES))
def test_transfer_matches_oracle(name, built_lib):
    dev = torch.device("cuda:0")
    sc = scenes.to_device(CASES[name]("cpu"), dev)
    pp, info, s = gpu_pipeline(sc)
    o = oracle_for_scene(CASES[name]("cpu"))
    o.prepass()
    o.build_regular_indices()
    assert info.n_regular == o.regular_count
    for a in range(3):
        assert np.array_equal(pp.regular_index(a), o.regular_index(a)), a
    s.solve(1e-10, 5000)
    x = s.solution()
    out = s.transfer_to_regular_grid()
    # oracle transfer fed with the SAME solution vector: everything must agree bit for bit
    want = o.transfer_to_regular_grid(x)
    for l in range(o.levels):
        lab_g, v_g = s.node_grid(l)
        lab_o, v_o = o.node_grid(l)
        assert np.array_equal(lab_g, lab_o), l
        for a in range(3):
            assert np.array_equal(v_g[a], v_o[a]), (l, a)
    for a in range(3):
        assert np.array_equal(out[a], want[a]), a
    # and end to end (oracle's own solve): the regular-grid velocity field within the north-star tolerance
    o.hot_path()
    xo, _ = o.solve(1e-10, 5000)
    ref = o.transfer_to_regular_grid(xo)
    g = np.concatenate([v.ravel() for v in out]).astype(np.float64)
    r = np.concatenate([v.ravel() for v in ref]).astype(np.float64)
    assert rel_l2(g, r) < 1e-5


def test_rigid_translation_survives_the_transfer(built_lib):
    dev = torch.device("cuda:0")
    sc = scenes.fat_beam(128, 4, device=dev)
    cv = (0.5, -2.0, 1.25)
    sc.velocity = scenes.constant_velocity(sc.res, cv, device=dev)
    pp, info, s = gpu_pipeline(sc)
    sinfo = s.solve(1e-8, 50)
    assert sinfo.iterations == 0
    out = s.transfer_to_regular_grid()
    for a in range(3):
        assert np.array_equal(out[a], np.full_like(out[a], cv[a]))   # interpolation weights sum to one, exactly


def test_transfer_full_size_properties(built_lib):
    """512^3: only the regular DOF faces change, values stay within the range of the octree solution."""
    dev = torch.device("cuda:0")
    sc = scenes.fat_beam(512, 4, device=dev)
    pp, info, s = gpu_pipeline(sc)
    s.solve(1e-3, 2500)
    x = s.solution()
    out = s.transfer_to_regular_grid()
    for a in range(3):
        vin = sc.velocity[a].cpu().numpy()
        ri = pp.regular_index(a)
        changed = out[a] != vin
        assert not changed[ri == capi.UNASSIGNED].any()
        assert np.isfinite(out[a]).all()
        sel = ri >= 0
        assert out[a][sel].min() >= x.min() - 1e-3 and out[a][sel].max() <= x.max() + 1e-3
