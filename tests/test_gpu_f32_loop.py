"""The float-vector PCG loop of AVS_PRECISION_F32 contexts (csrc/avs_pcg_f32.inl; the reference built with USESINGLEPRECISION,
/root/reference/Source/HDK_Utilities.h:25-37: SolveType = fpreal32 through Eigen::ConjugateGradient, HDK_AdaptiveViscosity.cpp:613-630).
* the float SpMV kernels (k_spmv_brick<.., float>, its value-code variant, the streaming kernel for matrices without the form) give, for a
  float x, exactly the float row sums -- one multiply and one add per entry, left to right in the stored column order -- that the
  oracle's SPMV_F defines: bit for bit against a numpy float32 restatement on the ORACLE's f32 system;
* the loop's iteration count agrees with the oracle's float CG (orc_pcg_csr_f32) to a few per cent and its solution to float-CG accuracy
  (neither side reproduces Eigen's vectorised reduction order: the dots differ in the last float bits);
* AVS_OPTION_F32_VECTORS = 0 restores the fp64 iteration on the float system."""
import ctypes as C

import numpy as np
import pytest
import torch

from adaptiveviscositysolver_amd import ViscositySolve, capi, scenes
from util import build_pyramid, feed, oracle_from_pyramid, rel_l2

pytestmark = pytest.mark.gpu

SCENES = {
    "beam128_L4": (lambda: scenes.fat_beam(128, 4), True),                                     # brick form, one dictionary
    "sheet128_L4": (lambda: scenes.thin_sheet(128, 4, thickness_cells=12), True),
    "beam64_L3_wall": (lambda: scenes.fat_beam(64, 3, wall=True), True),
    "beam128_L4_varvisc": (lambda: scenes.fat_beam(128, 4, variable_viscosity=True), True),    # brick form, value-code variant
    "beam64_L3_stream": (lambda: scenes.fat_beam(64, 3), False),                               # no form: the streaming kernel, one dictionary
    "sphere32_L3_stream": (lambda: scenes.sphere(32, 3, radius=0.36), False),                  # ... thousands of values
    "beam32_varvisc_stream": (lambda: scenes.fat_beam(32, 2, wall=True, variable_viscosity=True), False),
}


def _solver(sc, monkeypatch, brick, probe=True):
    monkeypatch.setenv("AVS_BRICK", "1" if brick else "0")
    pyr = build_pyramid(sc)
    dsc = scenes.to_device(sc, torch.device("cuda:0"))
    s = ViscositySolve(sc.res, sc.dx, sc.dt, pyr.levels, device=0, probe=probe, precision=capi.PRECISION_F32)
    feed(s, pyr)
    s.set_scene_fields(dsc)
    s.set_solver_option(capi.OPTION_F32_VECTORS, 1)     # always (the default, -1, leaves systems that fit the chip to the CU-resident fp64 loop)
    return s, pyr


def _oracle_f32(sc, pyr):
    o = oracle_from_pyramid(sc, pyr)
    o.L.orc_set_precision(o.h, 1)
    o.hot_path()
    return o


def _float_row_sums(row_ptr, col, val, x):
    """SPMV_F of the oracle (oracle/avs_oracle.c, orc_pcg_csr_f32): s = 0.f; s += v[k] * x[col[k]] for k in the row's stored order."""
    rp = np.asarray(row_ptr, dtype=np.int64)
    v = val.astype(np.float32)
    xf = x.astype(np.float32)
    n = len(rp) - 1
    length = rp[1:] - rp[:-1]
    s = np.zeros(n, dtype=np.float32)
    for j in range(int(length.max()) if n else 0):
        m = length > j
        k = rp[:-1][m] + j
        s[m] = s[m] + v[k] * xf[col[k]]          # float32 multiply, float32 add: one rounding each
    return s


@pytest.mark.parametrize("name", list(SCENES))
def test_f32_product_is_the_float_row_sum(name, monkeypatch, built_lib):
    make, brick = SCENES[name]
    sc = make()
    s, pyr = _solver(sc, monkeypatch, brick)
    ai = s.assemble()
    fmt = s.matrix_format()
    if brick:
        assert fmt.brick_tiles > 0 and fmt.brick_pattern_rows >= 0.5 * ai.n_velocity, "the brick form did not run"
        assert fmt.brick_value_codes == (1 if "varvisc" in name else 0)
    else:
        assert fmt.brick_tiles == 0
    o = _oracle_f32(sc, pyr)
    A = o.csr()
    n = int(ai.n_velocity)
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(5)
    for trial in range(2):
        x = (rng.standard_normal(n) * (10.0 ** rng.integers(-3, 4, n))).astype(np.float32).astype(np.float64)
        want = _float_row_sums(A.row_ptr, A.col, A.val, x)
        dx = torch.from_numpy(x).to(dev)
        for fused in (0, 1):
            dy = torch.full((n,), float("nan"), dtype=torch.float64, device=dev)
            dot = C.c_double()
            capi.check(s.lib.avs_spmv_solver_form(s.h, dx.data_ptr(), dy.data_ptr(), fused, C.byref(dot)))
            got = dy.cpu().numpy()
            assert np.array_equal(got, got.astype(np.float32).astype(np.float64))
            gf = got.astype(np.float32)
            assert np.array_equal(gf.view(np.int32), want.view(np.int32)), (name, fused, int((gf != want).sum()))
            if fused:
                terms = x * want.astype(np.float64)
                assert abs(dot.value - float(terms.sum())) <= 2e-5 * max(1.0, float(np.abs(terms).sum()))
    s.close()


@pytest.mark.parametrize("name", ["beam128_L4", "beam128_L4_varvisc", "beam64_L3_stream", "sphere32_L3_stream"])
def test_f32_loop_against_the_oracle_float_cg(name, monkeypatch, built_lib):
    make, brick = SCENES[name]
    sc = make()
    s, pyr = _solver(sc, monkeypatch, brick, probe=False)
    s.assemble()
    tol = 1e-5
    info = s.solve(tol, 5000)
    x = s.solution()
    assert info.converged == 1 and info.resident == 0
    assert np.array_equal(x, x.astype(np.float32).astype(np.float64))          # Eigen::VectorXf
    o = _oracle_f32(sc, pyr)
    xo, io = o.solve(tol, 5000)
    assert abs(info.iterations - io.iterations) <= max(2, io.iterations // 50), (info.iterations, io.iterations)   # within 2 %
    # a second solve in the same context: same count, same bits (fixed reduction order)
    again = s.solve(tol, 5000)
    assert again.iterations == info.iterations and np.array_equal(s.solution(), x)
    # the yardstick: the fp64 iteration on the same float system (AVS_OPTION_F32_VECTORS = 0), converged far below float accuracy
    s.set_solver_option(capi.OPTION_F32_VECTORS, 0)
    i64 = s.solve(1e-9, 8000)
    x64 = s.solution()
    e_gpu, e_orc, d = rel_l2(x, x64), rel_l2(xo, x64), rel_l2(x, xo)
    print(f"{name}: iterations {info.iterations} (oracle float CG {io.iterations}, fp64 loop {i64.iterations} at 1e-9); "
          f"error vs the converged solution: float loop {e_gpu:.2e}, oracle float CG {e_orc:.2e}; float loop vs oracle {d:.2e}")
    assert i64.converged == 1
    assert e_gpu < max(1.5 * e_orc, 5e-5), (e_gpu, e_orc)        # as close to the float system's solution as Eigen-in-float gets
    assert d < 5e-5 + 2 * e_orc, (d, e_gpu, e_orc)
    s.close()


def test_f32_auto_mode_prefers_the_resident_loop_for_small_systems(monkeypatch, built_lib):
    """AVS_OPTION_F32_VECTORS = -1 (default): a float system that fits the chip runs the CU-resident fp64 loop (faster there); one that does
    not runs the float-vector loop"""
    sc = scenes.fat_beam(64, 3)
    s, pyr = _solver(sc, monkeypatch, False, probe=False)
    s.set_solver_option(capi.OPTION_F32_VECTORS, -1)
    s.assemble()
    info = s.solve(1e-5, 5000)
    assert info.converged == 1 and info.resident == 1
    x = s.solution()
    assert np.array_equal(x, x.astype(np.float32).astype(np.float64))
    s.set_solver_option(capi.OPTION_RESIDENT_LOOP, 0)      # no resident loop: auto falls to the float-vector loop
    i2 = s.solve(1e-5, 5000)
    assert i2.converged == 1 and i2.resident == 0 and rel_l2(s.solution(), x) < 5e-5
    s.close()


def test_f32_loop_tight_tolerance_terminates(monkeypatch, built_lib):
    """tol = 1e-10 is beyond what float CG resolves: the recursively updated residual still falls below the threshold (as in Eigen) or
    the iteration cap ends the solve; the answer stays the float system's solution"""
    sc = scenes.fat_beam(64, 3)
    s, pyr = _solver(sc, monkeypatch, False, probe=False)
    s.assemble()
    info = s.solve(1e-10, 2500)
    x = s.solution()
    assert np.all(np.isfinite(x)) and info.iterations <= 2500
    s.set_solver_option(capi.OPTION_F32_VECTORS, 0)
    s.solve(1e-10, 2500)
    assert rel_l2(x, s.solution()) < 2e-4
    s.close()


def test_f32_loop_plain_cg_option(monkeypatch, built_lib):
    """the float loop without a preconditioner (the build without USEEIGEN, cpp:638-642, in float): the inverse diagonal it multiplies with is
    1.f; iteration count and solution against the oracle's float CG in the same mode"""
    sc = scenes.fat_beam(64, 3, wall=True)
    s, pyr = _solver(sc, monkeypatch, False, probe=False)
    s.assemble()
    jac = s.solve(1e-5, 8000)
    s.set_solver_option(capi.OPTION_PRECONDITIONER, capi.PRECONDITIONER_NONE)
    info = s.solve(1e-5, 8000)
    x = s.solution()
    o = _oracle_f32(sc, pyr)
    o.L.orc_set_preconditioner(o.h, 1)
    xo, io = o.solve(1e-5, 8000)
    assert info.converged == 1 and info.resident == 0 and info.iterations > jac.iterations
    assert abs(info.iterations - io.iterations) <= max(3, io.iterations // 25), (info.iterations, io.iterations)
    assert rel_l2(x, xo) < 2e-4
    s.close()


def test_f32_loop_graph_replay_changes_nothing(monkeypatch, built_lib):
    """chunks replayed from the captured hipGraph or enqueued launch by launch: same iteration count, same solution bits"""
    sc = scenes.fat_beam(128, 4)
    s, pyr = _solver(sc, monkeypatch, True, probe=False)
    s.assemble()
    a = s.solve(1e-5, 5000)
    xa = s.solution()
    s.set_solver_option(capi.OPTION_GRAPH_REPLAY, 0)
    b = s.solve(1e-5, 5000)
    assert a.iterations == b.iterations > 64 and np.array_equal(xa, s.solution())
    s.close()


def test_f32_loop_can_be_cancelled(monkeypatch, built_lib):
    """avs_cancel (the reference's opInterrupt, cpp:2528) ends the float loop at its next poll like the fp64 loops"""
    import threading
    sc = scenes.fat_beam(128, 4)
    s, pyr = _solver(sc, monkeypatch, True, probe=False)
    s.assemble()
    capi.check(s.lib.avs_cancel(s.h))
    info = s.solve(1e-8, 5000)                      # a pending request cancels the next solve before its first iteration
    assert info.cancelled == 1 and info.converged == 0 and info.iterations == 0
    N = 1000                                        # (in float the recurrence residual reaches FLT_MIN -- "converged" -- well before the fp64 loops' ~2,300 iterations)
    full = s.solve(1e-30, N)
    assert full.iterations == N and not full.converged and not full.cancelled
    canceller = threading.Timer(0.3 * full.solve_ms * 1e-3, lambda: capi.check(s.lib.avs_cancel(s.h)))
    canceller.start()
    info = s.solve(1e-30, N)
    canceller.join()
    assert info.cancelled == 1 and info.converged == 0 and 0 < info.iterations < N
    again = s.solve(1e-5, 5000)
    assert again.converged == 1 and again.cancelled == 0
    s.close()
