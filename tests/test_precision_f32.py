"""SolveType = fpreal32 (the reference built with USESINGLEPRECISION, /root/reference/Source/HDK_Utilities.h:25-37; avs_desc.precision =
AVS_PRECISION_F32): every Eigen::Triplet<float> narrowed where it is built (cpp:2447, 2768), duplicates summed in float (cpp:613-614), the
right-hand side updated in float steps (cpp:2456, 2772), the initial guess narrowed at its store (cpp:2371).
CPU: the oracle's f32 mode produces float values, close to (and different from) the fp64 system; its float CG reaches the float system's
solution.  GPU: matrix, rhs and x0 bit-exact against the oracle's f32 mode, solution within the float CG's accuracy."""
import numpy as np
import pytest

from adaptiveviscositysolver_amd import scenes
from util import oracle_for_scene, rel_l2

SCENES = {
    "sphere32_L3": lambda: scenes.sphere(32, 3, radius=0.36),
    "beam32_wall_varvisc": lambda: scenes.fat_beam(32, 2, wall=True, variable_viscosity=True),
    "obstacle_rho_usolid": lambda: scenes.with_sampled_fields(scenes.sphere_with_obstacle(32, 3)),
}


def _oracle(sc, f32):
    o = oracle_for_scene(sc, f32=f32)
    o.prepass()
    o.hot_path()
    return o


@pytest.mark.parametrize("name", list(SCENES))
def test_oracle_f32_mode(name):
    sc = SCENES[name]()
    o64, o32 = _oracle(sc, False), _oracle(sc, True)
    A64, A32 = o64.csr(), o32.csr()
    assert np.array_equal(A64.row_ptr, A32.row_ptr) and np.array_equal(A64.col, A32.col)     # same structure
    for arr in (A32.val, A32.rhs, o32.initial_guess()):
        assert np.array_equal(arr, arr.astype(np.float32).astype(np.float64))               # float values in double arrays
    assert not np.array_equal(A64.val, A32.val)
    assert rel_l2(A32.val, A64.val) < 1e-7 and rel_l2(A32.rhs, A64.rhs) < 1e-6 and rel_l2(o32.initial_guess(), o64.initial_guess()) < 1e-7
    # not simply the fp64 system narrowed at the end: duplicates are summed in float, the rhs in float steps
    narrowed = A64.val.astype(np.float32).astype(np.float64)
    assert np.count_nonzero(narrowed != A32.val) > 0
    x64, _ = o64.solve(1e-4, 5000)
    x32, i32 = o32.solve(1e-4, 5000)
    assert np.array_equal(x32, x32.astype(np.float32).astype(np.float64))
    assert i32.iterations > 0 and rel_l2(x32, x64) < 5e-4


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(SCENES))
def test_gpu_f32_system_bit_exact_and_solve(name, built_lib):
    import torch
    from adaptiveviscositysolver_amd import ViscositySolve, capi
    from util import build_pyramid, feed, oracle_from_pyramid
    sc = SCENES[name]()
    dsc = scenes.to_device(sc, torch.device("cuda:0"))
    pyr = build_pyramid(dsc)
    s = ViscositySolve(sc.res, sc.dx, sc.dt, pyr.levels, device=0, precision=capi.PRECISION_F32)
    feed(s, pyr)
    s.set_scene_fields(dsc)
    s.assemble()
    o = oracle_from_pyramid(sc, pyr)
    o.L.orc_set_precision(o.h, 1)
    o.hot_path()
    rp, col, val, rhs = s.csr()
    A = o.csr()
    assert np.array_equal(rp, A.row_ptr) and np.array_equal(col, A.col)
    assert np.array_equal(val, A.val) and np.array_equal(rhs, A.rhs)
    assert np.array_equal(s.initial_guess(), o.initial_guess())
    assert np.array_equal(val, val.astype(np.float32).astype(np.float64))
    # the solve: fp64 iterations on the float system, solution narrowed to float; the oracle runs Eigen's algorithm in float
    info = s.solve(1e-5, 5000)
    x = s.solution()
    assert info.converged == 1 and np.array_equal(x, x.astype(np.float32).astype(np.float64))
    xo, io = o.solve(1e-5, 5000)
    assert rel_l2(x, xo) < 2e-4, (info.iterations, io.iterations)
    # and against the fp64 build of the same scene: the two precisions agree to float accuracy
    s64 = ViscositySolve(sc.res, sc.dx, sc.dt, pyr.levels, device=0)
    feed(s64, pyr)
    s64.set_scene_fields(dsc)
    s64.assemble()
    s64.solve(1e-8, 5000)
    assert rel_l2(x, s64.solution()) < 1e-4
    s.close()
    s64.close()


def test_oracle_plain_cg_option():
    """No preconditioner (the build without USEEIGEN passes nullptr to HDK's solveConjugateGradient, cpp:638-642): same solution,
    more iterations than Jacobi-PCG on a system whose diagonal varies (octree levels, wall)."""
    sc = scenes.fat_beam(32, 3, wall=True)
    o = _oracle(sc, False)
    xj, ij = o.solve(1e-8, 5000)
    o.L.orc_set_preconditioner(o.h, 1)
    xn, inn = o.solve(1e-8, 5000)
    assert inn.iterations > ij.iterations and rel_l2(xn, xj) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("name,resident", [("beam64_L3_wall", "1"), ("beam64_L3_wall", "0"), ("varvisc", "0")])
def test_gpu_plain_cg_option(name, resident, built_lib):
    import os
    import torch
    from adaptiveviscositysolver_amd import ViscositySolve, capi
    from util import build_pyramid, feed, oracle_from_pyramid
    sc = scenes.fat_beam(64, 3, wall=True) if name == "beam64_L3_wall" else scenes.fat_beam(32, 2, wall=True, variable_viscosity=True)
    dsc = scenes.to_device(sc, torch.device("cuda:0"))
    pyr = build_pyramid(dsc)
    s = ViscositySolve(sc.res, sc.dx, sc.dt, pyr.levels, device=0)
    feed(s, pyr)
    s.set_scene_fields(dsc)
    s.assemble()
    o = oracle_from_pyramid(sc, pyr)
    o.hot_path()
    old = os.environ.get("AVS_CG_RESIDENT")
    os.environ["AVS_CG_RESIDENT"] = resident
    s.set_solver_option(capi.OPTION_RESIDENT_LOOP, int(resident))
    try:
        jac = s.solve(1e-9, 8000)
        s.set_solver_option(capi.OPTION_PRECONDITIONER, capi.PRECONDITIONER_NONE)
        info = s.solve(1e-9, 8000)
        x = s.solution()
        o.L.orc_set_preconditioner(o.h, 1)
        xo, io = o.solve(1e-9, 8000)
        assert info.converged == 1 and info.iterations > jac.iterations
        assert abs(info.iterations - io.iterations) <= max(3, io.iterations // 50), (info.iterations, io.iterations)
        assert rel_l2(x, xo) < 1e-7
        s.set_solver_option(capi.OPTION_PRECONDITIONER, capi.PRECONDITIONER_JACOBI)     # and back
        again = s.solve(1e-9, 8000)
        assert abs(again.iterations - jac.iterations) <= 1
    finally:
        if old is None:
            os.environ.pop("AVS_CG_RESIDENT", None)
        else:
            os.environ["AVS_CG_RESIDENT"] = old
    s.close()
