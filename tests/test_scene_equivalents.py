"""Scene-equivalent workloads: the slender clamped beam of Scenes/viscousBeam.hip and the poured sheet of
Scenes/viscousBuckling.hip, with the scene files' parameters (dx, box, viscosity, density, dt, levels, colliders) on NON-power-of-two
simulation grids; buckling also has the non-dyadic fp32 dx = (double)(float)1e-3.

CPU: the oracle on coarsened copies (plumbing + the independent scatter-form check).  GPU: the full-resolution scenes through the
product path -- device pre-pass and solve context created with field_n* = the simulation grid, fields cropped to it, as the HDK shim
hands them over -- against the oracle on the padded octree lattice: pyramid, CSR, rhs, x0 bit-exact, solution to 1e-5."""
import numpy as np
import pytest

from adaptiveviscositysolver_amd import scenes
from independent import check_scatter_form
from oracle import oracle as O
from util import oracle_for_scene, rel_l2

SCENES = {"beam": scenes.viscous_beam_scene, "buckling": scenes.viscous_buckling_scene}


@pytest.mark.parametrize("name,coarsen", [("beam", 4), ("beam", 2), ("buckling", 2)])
def test_oracle_on_coarsened_scene(name, coarsen):
    sc = SCENES[name](coarsen=coarsen)
    assert any(r & (r - 1) for r in sc.field_res), "the simulation grid is meant to be non-power-of-two"
    assert all(p >= f and p & (p - 1) == 0 for p, f in zip(sc.res, sc.field_res))
    o = oracle_for_scene(sc)
    o.prepass()
    o.hot_path()
    A = o.csr()
    es, cs = o.edge_stencils(), o.center_stencils()
    assert A.n > 10000 and int((es["bcnt"] > 0).sum()) > 100          # the collider produces boundary terms
    if name == "beam":
        assert o.levels >= (4 if coarsen == 2 else 2)
    r = check_scatter_form(A.row_ptr, A.col, A.val, A.rhs, o.initial_guess(), es, cs, o.count(O.I_CENTER))
    assert r["dups"] == 0 and r["mass_min"] >= 0.0
    x, info = o.solve(1e-3, 2500)                                      # the scenes' own tolerance / iteration cap
    assert info.error <= 1e-3 and 10 < info.iterations < 2500


def test_buckling_dx_is_the_fp32_voxel_size():
    sc = scenes.viscous_buckling_scene(coarsen=2)
    assert sc.dx == 2 * float(np.float32(1e-3)) and sc.dx != 2e-3


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["beam", "buckling"])
def test_scene_equivalent_matches_oracle(name, built_lib):
    import torch
    from adaptiveviscositysolver_amd import DevicePrepass, ViscositySolve, capi
    sc = SCENES[name]()
    o = oracle_for_scene(sc)
    o.prepass()
    o.hot_path()
    dsc = scenes.to_device(scenes.crop_to_field(sc), torch.device("cuda:0"))
    pp = DevicePrepass(sc.res, sc.dx, sc.levels, field_res=sc.field_res)
    info = pp.run(dsc.liquid, dsc.solid)
    assert info.levels == o.levels and (info.n_velocity, info.n_edge, info.n_center) == (o.count(0), o.count(1), o.count(2))
    for l in range(o.levels):
        assert np.array_equal(pp.labels(l), o.labels(l)), l
        for a in range(3):
            assert np.array_equal(pp.index(capi.INDEX_VELOCITY, l, a), o.index(O.I_VELOCITY, l, a)), (l, a)
            assert np.array_equal(pp.index(capi.INDEX_EDGE, l, a), o.index(O.I_EDGE, l, a)), (l, a)
        assert np.array_equal(pp.index(capi.INDEX_CENTER, l), o.index(O.I_CENTER, l)), l
    s = ViscositySolve(sc.res, sc.dx, sc.dt, info.levels, device=0, field_res=sc.field_res)
    pp.apply(s)
    s.set_scene_fields(dsc)
    s.assemble()
    rp, col, val, rhs = s.csr()
    A = o.csr()
    assert np.array_equal(rp, A.row_ptr) and np.array_equal(col, A.col)
    assert np.array_equal(val, A.val) and np.array_equal(rhs, A.rhs)
    assert np.array_equal(s.initial_guess(), o.initial_guess())
    es, eo = s.edge_stencils(), o.edge_stencils()
    assert np.array_equal(es["bval"], eo["bval"]) and np.array_equal(es["weight"], eo["weight"])
    got = s.solve(1e-10, 6000)
    xo, io = o.solve(1e-10, 6000)
    assert got.converged == 1 and abs(got.iterations - io.iterations) <= max(3, io.iterations // 100)
    assert rel_l2(s.solution(), xo) < 1e-5                              # north_star tolerance
    g3, (_, o3) = s.solve(1e-3, 2500), o.solve(1e-3, 2500)              # the scene's own settings (cpp:63, 66)
    assert abs(g3.iterations - o3.iterations) <= 3
    s.close()
    pp.close()
