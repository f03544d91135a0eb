"""GPU parity tests proper: HIP path (through the C ABI) vs the CPU oracle on identical inputs.

Bars (BASELINE.json north_star): integer work (DOF ids, CSR pattern, stencil indices) bit-exact;
velocity field within 1e-5 relative L2 with both solvers run to a tight tolerance.  Because the
assembly kernels follow the reference's operation order with one rounding per operation, the
fp64 stencil coefficients, weights, CSR values, rhs and initial guess are ALSO required bit-exact.
"""
import numpy as np
import pytest
import torch

from adaptiveviscositysolver_amd import ViscositySolve, capi, pcg_csr, scenes
from util import build_pyramid, feed, oracle_for_scene, oracle_from_pyramid, rel_l2
from oracle import oracle as O

pytestmark = pytest.mark.gpu

SCENES = {
    "beam32": lambda dev: scenes.fat_beam(32, 3, device=dev),
    "beam64_L2": lambda dev: scenes.fat_beam(64, 2, device=dev),              # BASELINE configs[0]
    "beam64_L3_wall": lambda dev: scenes.fat_beam(64, 3, wall=True, device=dev),
    "beam64_varvisc": lambda dev: scenes.fat_beam(64, 4, variable_viscosity=True, device=dev),
    "sphere64": lambda dev: scenes.sphere(64, 4, device=dev),
    "sheet64": lambda dev: scenes.thin_sheet(64, 3, thickness_cells=12, device=dev),
    "beam_noncubic": lambda dev: scenes.fat_beam(64, 3, res=(64, 32, 32), device=dev),
    # five levels really appear (level-4 cells in the core): 12^4-leaf restriction rows (several batches of 144-leaf units per
    # lane group), level-4 rows in the row sweep, rows with > 64 raw triplets next to the coarsest cells
    "beam256_L5": lambda dev: scenes.fat_beam(256, 5, device=dev),
    # centre-lattice density TENSOR (cpp:2759-2766) + spatially varying solid velocity (cpp:1896-1905, 1952-1960): branches no
    # constant-field scene reaches (round-2 review, weak #2)
    "beam64_wall_rho_usolid": lambda dev: scenes.with_sampled_fields(scenes.fat_beam(64, 3, wall=True, device=dev)),
    "sphere64_obstacle_rho_usolid": lambda dev: scenes.with_sampled_fields(scenes.sphere_with_obstacle(64, 4, device=dev)),
    # an open TANK: liquid on the domain border on five sides, the collision SDF on the walls and the floor (round-3 review, item 9b):
    # border faces, ghost faces towards the solid (cpp:1757-1762, 1201-1320), on a power-of-two grid (no padding convention involved)
    "tank64_L3": lambda dev: scenes.tank(64, 3, device=dev),
    "tank128_L4_usolid": lambda dev: scenes.with_sampled_fields(scenes.tank(128, 4, device=dev)),
}


def gpu_solve_for(sc, pyr, enhanced=True):
    s = ViscositySolve(sc.res, sc.dx, sc.dt, pyr.levels, use_enhanced_gradients=enhanced, device=0)
    feed(s, pyr)
    s.set_scene_fields(sc)
    return s


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.mark.parametrize("name", list(SCENES))
def test_assembly_bit_exact(name, dev, built_lib):
    sc = scenes.to_device(SCENES[name]("cpu"), dev)
    pyr = build_pyramid(sc)
    s = gpu_solve_for(sc, pyr)
    ai = s.assemble()
    sc_h = SCENES[name]("cpu")
    o = oracle_for_scene(sc_h)
    o.prepass()
    # the pre-pass run on the GPU (avs_prepass.hip) must reproduce the oracle's integer pyramids
    assert pyr.levels == o.levels
    assert (pyr.n_velocity, pyr.n_edge, pyr.n_center) == (o.count(0), o.count(1), o.count(2))
    for l in range(o.levels):
        assert np.array_equal(pyr.labels[l], o.labels(l))
        for a in range(3):
            assert np.array_equal(pyr.vidx[l][a], o.index(O.I_VELOCITY, l, a))
            assert np.array_equal(pyr.eidx[l][a], o.index(O.I_EDGE, l, a))
        assert np.array_equal(pyr.cidx[l], o.index(O.I_CENTER, l))
    o.hot_path()
    # stencils
    for got, want in ((s.edge_stencils(), o.edge_stencils()), (s.center_stencils(), o.center_stencils())):
        for k in ("cnt", "idx", "bcnt"):
            assert np.array_equal(got[k], want[k]), k
        for k in ("coef", "bval", "weight"):
            assert np.array_equal(got[k], want[k]), k
    assert np.array_equal(s.initial_guess(), o.initial_guess())
    rp, col, val, rhs = s.csr()
    A = o.csr()
    assert ai.n_velocity == A.n and ai.nnz == len(A.col) and ai.raw_triplets == o.raw_triplets
    if name == "beam256_L5":
        assert o.levels == 5 and int((o.dof_table(0)[:, 0] & 0xff).max()) == 4   # level-4 velocity DOFs exist
    assert np.array_equal(rp, A.row_ptr.astype(np.int32))
    assert np.array_equal(col, A.col)
    assert np.array_equal(val, A.val)
    assert np.array_equal(rhs, A.rhs)


@pytest.mark.parametrize("name", ["beam32", "beam64_L3_wall", "beam64_varvisc", "sphere64", "sphere64_obstacle_rho_usolid"])
def test_solve_matches_oracle(name, dev, built_lib):
    sc = scenes.to_device(SCENES[name]("cpu"), dev)
    pyr = build_pyramid(sc)
    s = gpu_solve_for(sc, pyr)
    s.assemble()
    tol = 1e-10  # tight: at 1e-3 two correct CGs may differ by 1e-3 (SURVEY 7 "hard parts")
    info = s.solve(tol, 5000)
    x = s.solution()
    o = oracle_from_pyramid(SCENES[name]("cpu"), pyr)
    o.hot_path()
    xo, io = o.solve(tol, 5000)
    assert info.converged == 1
    assert abs(info.iterations - io.iterations) <= 3, (info.iterations, io.iterations)
    assert rel_l2(x, xo) < 1e-5          # north_star tolerance
    assert info.error <= tol
    # default tolerance run: iteration counts agree within a few
    info3 = s.solve(1e-3, 2500)
    _, io3 = o.solve(1e-3, 2500)
    assert abs(info3.iterations - io3.iterations) <= 3, (info3.iterations, io3.iterations)


def test_enhanced_gradients_off(dev, built_lib):
    sc = scenes.sphere(64, 4, device=dev)
    sc.use_enhanced_gradients = False
    pyr = build_pyramid(sc)
    s = gpu_solve_for(sc, pyr, enhanced=False)
    s.assemble()
    sc_h = scenes.sphere(64, 4)
    sc_h.use_enhanced_gradients = False
    o = oracle_from_pyramid(sc_h, pyr)
    o.hot_path()
    rp, col, val, rhs = s.csr()
    A = o.csr()
    assert np.array_equal(rp, A.row_ptr.astype(np.int32)) and np.array_equal(col, A.col)
    assert np.array_equal(val, A.val) and np.array_equal(rhs, A.rhs)


def test_rigid_translation_is_a_fixed_point(dev, built_lib):
    """D u = 0 for a rigid translation => A u = M u, b = M u, x0 = u: zero iterations."""
    sc = scenes.sphere(64, 4, device=dev)
    sc.velocity = scenes.constant_velocity(sc.res, (0.25, -1.5, 0.75), device=dev)
    pyr = build_pyramid(sc)
    s = gpu_solve_for(sc, pyr)
    s.assemble()
    info = s.solve(1e-8, 100)
    assert info.iterations == 0 and info.converged == 1
    x = s.solution()
    assert np.array_equal(x, s.initial_guess())


@pytest.mark.parametrize("variant", [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24])
def test_spmv_variants(variant, dev, built_lib):
    sc = scenes.fat_beam(64, 3, device=dev)
    pyr = build_pyramid(sc)
    s = gpu_solve_for(sc, pyr)
    s.assemble()
    rp, col, val, rhs = s.csr()
    n = len(rhs)
    rng = np.random.default_rng(7)
    x = rng.standard_normal(n)
    want = O.spmv_csr(rp.astype(np.int64), col, val, x)
    lib = capi.load_probe()      # avs_spmv_csr: a measurement entry (include/avs_probe.h)
    t = lambda a: torch.from_numpy(a).to(dev)
    d_rp, d_col, d_val, d_x = t(rp), t(col), t(val), t(x)
    d_y = torch.zeros(n, dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    capi.check(lib.avs_spmv_csr(n, d_rp.data_ptr(), d_col.data_ptr(), d_val.data_ptr(), d_x.data_ptr(),
                                d_y.data_ptr(), variant, 1, stream))
    torch.cuda.synchronize()
    got = d_y.cpu().numpy()
    if variant not in (2, 3, 4):   # stream kernels sums each row left to right like the oracle: bit-exact
        assert np.array_equal(got, want)
    else:
        assert np.allclose(got, want, rtol=1e-13, atol=1e-9 * np.abs(want).max())


def test_seam_a_pcg_csr_host_arrays(dev, built_lib):
    sc = scenes.fat_beam(32, 3)
    o = oracle_for_scene(sc)
    o.prepass()
    o.hot_path()
    A = o.csr()
    x0 = o.initial_guess()
    x, info = pcg_csr(A.row_ptr, A.col, A.val, A.rhs, x0, 1e-10, 5000)
    xo, io = o.solve(1e-10, 5000)
    assert info.converged == 1 and abs(info.iterations - io.iterations) <= 3
    assert rel_l2(x, xo) < 1e-5
    # zero right-hand side => x = 0, 0 iterations (Eigen: x.setZero())
    xz, iz = pcg_csr(A.row_ptr, A.col, A.val, np.zeros_like(A.rhs), x0, 1e-3, 10)
    assert iz.iterations == 0 and not xz.any()


def test_error_paths(dev, built_lib):
    with pytest.raises(capi.AvsError) as e:
        ViscositySolve((48, 32, 32), 1 / 32, 0.01, 2)
    assert e.value.status == capi.EINVAL
    s = ViscositySolve((32, 32, 32), 1 / 32, 0.01, 2)
    with pytest.raises(capi.AvsError) as e:
        s.solve()
    assert e.value.status == capi.ESTATE
    with pytest.raises(capi.AvsError) as e:
        s.assemble()
    assert e.value.status == capi.ESTATE


def test_empty_domain_has_no_dofs(dev, built_lib):
    """No liquid at all (the reference asserts cappedLevel != 0, oct.cpp:206, so the pre-pass yields no
    levels): a 1-level pyramid of INACTIVE cells / UNASSIGNED indices must assemble and solve as no-ops."""
    n = 16
    sc = scenes.fat_beam(n, 1, device=dev)
    sc.liquid = torch.full_like(sc.liquid, 10.0)
    assert build_pyramid(sc).levels == 0
    s = ViscositySolve((n, n, n), 1.0 / n, 0.01, 1, device=0)
    s.set_labels(0, torch.zeros((n, n, n), dtype=torch.int8, device=dev))
    for a in range(3):
        fs = [n, n, n]
        fs[2 - a] += 1
        es = [n + 1, n + 1, n + 1]
        es[2 - a] -= 1
        s.set_index_field(capi.INDEX_VELOCITY, 0, a, torch.full(fs, -1, dtype=torch.int32, device=dev))
        s.set_index_field(capi.INDEX_EDGE, 0, a, torch.full(es, -1, dtype=torch.int32, device=dev))
    s.set_index_field(capi.INDEX_CENTER, 0, 0, torch.full((n, n, n), -1, dtype=torch.int32, device=dev))
    s.set_dof_counts(0, 0, 0)
    ai = s.assemble()
    assert ai.n_velocity == 0 and ai.nnz == 0
    info = s.solve(1e-3, 10)
    assert info.iterations == 0
    assert len(s.solution()) == 0


def test_single_level_uniform_known_answer(dev, built_lib):
    """1-level tree, liquid everywhere: the GPU matrix row equals the closed form of SURVEY A.8."""
    n = 16
    liquid = torch.full((n, n, n), -100.0, dtype=torch.float32, device=dev)
    sc = scenes.Scene(res=(n, n, n), dx=1.0 / n, dt=0.5, levels=1, liquid=liquid, viscosity=8.0, density=4.0,
                      velocity=scenes.smooth_velocity((n, n, n), 1.0 / n, device=dev))
    pyr = build_pyramid(sc)
    assert pyr.levels == 1
    s = gpu_solve_for(sc, pyr)
    s.assemble()
    rp, col, val, rhs = s.csr()
    kappa = sc.dt * sc.viscosity / sc.dx ** 2
    row = int(pyr.vidx[0][0][n // 2, n // 2, n // 2])
    v = val[rp[row]:rp[row + 1]]
    c = col[rp[row]:rp[row + 1]]
    assert len(v) == 15 and v[c == row][0] == sc.density + 8 * kappa
    assert sorted(np.round(v[c != row] / kappa).astype(int).tolist()) == [-2, -2] + [-1] * 8 + [1] * 4
    assert rhs[row] == sc.density * s.initial_guess()[row]


def test_inconsistent_inputs_are_rejected(dev, built_lib):
    sc = scenes.fat_beam(32, 3, device=dev)
    pyr = build_pyramid(sc)
    s = gpu_solve_for(sc, pyr)
    s.set_dof_counts(pyr.n_velocity - 5, pyr.n_edge, pyr.n_center)     # an id >= declared count
    with pytest.raises(capi.AvsError) as e:
        s.assemble()
    assert e.value.status == capi.EINVAL
    s.set_dof_counts(pyr.n_velocity + 5, pyr.n_edge, pyr.n_center)     # ids missing
    with pytest.raises(capi.AvsError) as e:
        s.assemble()
    assert e.value.status == capi.EINVAL
    s.set_dof_counts(pyr.n_velocity, pyr.n_edge, pyr.n_center)
    # corrupt one velocity index so that a stencil no longer contains its row DOF: reference assert -> AVS_EINTERNAL
    bad = pyr.vidx[0][0].copy()
    k = tuple(np.argwhere(bad >= 0)[100])
    k2 = tuple(np.argwhere(bad >= 0)[5000])
    a, b = int(bad[k]), int(bad[k2])
    bad[k], bad[k2] = b, a
    s.set_index_field(capi.INDEX_VELOCITY, 0, 0, np.ascontiguousarray(bad))
    s.assemble()   # still a consistent numbering (a permutation): must assemble
    assert s.info().n_velocity == pyr.n_velocity


def test_fields_on_a_smaller_simulation_grid(dev, built_lib):
    """HDK_OctreeGrid::init stretches the octree grid to powers of two (oct.cpp:13-24) while the scalar fields stay on
    the simulation grid; cells outside it are INACTIVE (oct.cpp:375-379).  With avs_desc.field_n* the ABI takes the
    fields at their own resolution.  Reference here: the oracle on fields given on the full 64^3 lattice -- whatever
    lies outside the 48 x 40 x 24 simulation grid must not matter, so CSR, rhs, initial guess are bit-identical."""
    n, field_res = 64, (48, 40, 24)
    dx = 1.0 / n
    liquid = scenes.box_sdf((n, n, n), dx, center=(24 * dx, 20 * dx, 12 * dx), half=(17 * dx, 13 * dx, 6.5 * dx))
    x = (torch.arange(n, dtype=torch.float64) + 0.5) * dx
    visc = (150.0 * (1.0 + 5.0 * x))[None, None, :].expand(n, n, n).to(torch.float32).contiguous()
    sc = scenes.Scene(res=(n, n, n), dx=dx, dt=1.0 / 60.0, levels=3, liquid=liquid, viscosity=visc, density=900.0,
                      velocity=scenes.smooth_velocity((n, n, n), dx, gravity_dt=0.1), name="corner_box")
    import prepass_torch
    pyr = prepass_torch.build_pyramid(sc)          # pyramid on the padded 64^3 lattice (host tensors)
    o = oracle_from_pyramid(sc, pyr)
    o.hot_path()
    want = o.csr()
    s = ViscositySolve(sc.res, sc.dx, sc.dt, pyr.levels, device=0, field_res=field_res)
    feed(s, pyr)                                  # index pyramids at 64^3; weight fields cropped by set_field
    s.set_scene_fields(scenes.to_device(sc, dev))       # viscosity / velocity arrays cropped likewise (device path)
    s.assemble()
    rp, col, val, rhs = s.csr()
    assert np.array_equal(rp, want.row_ptr) and np.array_equal(col, want.col)
    assert np.array_equal(val, want.val) and np.array_equal(rhs, want.rhs)
    assert np.array_equal(s.initial_guess(), o.initial_guess())
    info = s.solve(1e-10, 4000)
    xo, oi = o.solve(1e-10, 4000)
    assert info.converged == 1 and abs(info.iterations - oi.iterations) <= 2
    assert rel_l2(s.solution(), xo) < 1e-8
    s.close()
    with pytest.raises(capi.AvsError):                  # a simulation grid larger than the octree grid is rejected
        ViscositySolve(sc.res, sc.dx, sc.dt, pyr.levels, device=0, field_res=(65, 64, 64))


@pytest.mark.parametrize("graph", ["1", "0"])
def test_solve_is_deterministic(graph, dev, built_lib, monkeypatch):
    """All reductions use fixed orders (block partials folded in index order, multi-block fold in block order, DPP wave
    sums): repeating a solve -- with or without hipGraph replay -- reproduces the solution bit for bit."""
    monkeypatch.setenv("AVS_PCG_GRAPH", graph)
    sc = scenes.fat_beam(128, 3, device=dev)          # 380 k rows: multi-block reduction and graph replay both active
    pyr = build_pyramid(sc)
    s = gpu_solve_for(sc, pyr)
    s.assemble()
    runs = []
    for _ in range(3):
        info = s.solve(1e-8, 5000)
        runs.append((info.iterations, info.error, s.solution()))
    for it, err, x in runs[1:]:
        assert it == runs[0][0] and err == runs[0][1]
        assert np.array_equal(x, runs[0][2])
    s.close()


@pytest.mark.parametrize("n0", [64, 128])
def test_fused_scalar_steps_match_separate_reductions(n0, dev, built_lib, monkeypatch):
    """Single-GPU loop: the beta step folded into k_update_xp (and, for <= 4096 SpMV partials -- the 64^3 case --, the alpha
    step folded into k_update_r) against the loop with a k_reduce launch per step.  Same algorithm, different summation
    trees: same iteration count +-1 and the same solution to rounding; iteration caps that stop in an odd / even
    iteration exercise the parity-slotted r.z; repeated and interleaved solves on one context must not leak state."""
    sc = scenes.fat_beam(n0, 3, device=dev)
    pyr = build_pyramid(sc)
    s = gpu_solve_for(sc, pyr)
    s.assemble()
    ref = {}
    for mode in ("0", "1", "0", "1"):
        s.set_solver_option(capi.OPTION_FUSED_SCALAR_STEPS, int(mode))
        info = s.solve(1e-9, 5000)
        x = s.solution()
        assert info.converged == 1
        if mode in ref:                                  # same mode again: bit for bit (fixed summation orders)
            assert info.iterations == ref[mode][0] and np.array_equal(x, ref[mode][1])
        ref[mode] = (info.iterations, x)
    assert abs(ref["1"][0] - ref["0"][0]) <= 1
    assert rel_l2(ref["1"][1], ref["0"][1]) < 1e-8
    for cap in (1, 2, 31, 32, 33, 64, 65):               # stop by the cap: x after exactly `cap` iterations in both loops
        xs = {}
        for mode in ("1", "0"):
            s.set_solver_option(capi.OPTION_FUSED_SCALAR_STEPS, int(mode))
            info = s.solve(1e-12, cap)
            assert info.iterations == cap and info.converged == 0
            xs[mode] = s.solution()
        assert rel_l2(xs["1"], xs["0"]) < 1e-10, cap
    s.close()


def test_graph_replay_equals_plain_launches(dev, built_lib, monkeypatch):
    """Replaying captured hipGraph chunks runs the same kernels on the same data: identical iterations and solution."""
    sc = scenes.fat_beam(64, 3, device=dev)
    pyr = build_pyramid(sc)
    s = gpu_solve_for(sc, pyr)
    s.assemble()
    out = {}
    for mode in ("1", "0"):
        s.set_solver_option(capi.OPTION_GRAPH_REPLAY, int(mode))
        info = s.solve(1e-9, 5000)
        out[mode] = (info.iterations, s.solution())
    assert out["1"][0] == out["0"][0] and out["1"][0] > 64      # several chunks: the graph really was replayed
    assert np.array_equal(out["1"][1], out["0"][1])
    s.close()


def test_state_the_reference_asserts_on_is_rejected(dev, built_lib):
    """Same inputs as tests/test_oracle_known_answers.py::test_state_the_reference_asserts_on_is_rejected_not_crashed:
    the device path reports AVS_EINTERNAL (addError + return false in the plugin), it neither crashes nor reads out
    of bounds."""
    from test_oracle_known_answers import leaving_box_scene
    from adaptiveviscositysolver_amd import DevicePrepass
    sc = leaving_box_scene()
    dsc = scenes.to_device(sc, dev)
    pp = DevicePrepass(sc.res, sc.dx, sc.levels)
    pi = pp.run(dsc.liquid, dsc.solid)
    s = ViscositySolve(sc.res, sc.dx, sc.dt, pi.levels, device=0)
    pp.apply(s)
    s.set_scene_fields(dsc)
    with pytest.raises(capi.AvsError) as e:
        s.assemble()
    assert e.value.status == capi.EINTERNAL
    pp.close()
    s.close()


@pytest.mark.parametrize("n", [1, 2, 3, 5, 63, 64, 65, 513])
def test_seam_a_tiny_systems(n, dev, built_lib):
    """avs_pcg_csr on very small SPD systems (tridiagonal): the value-indexed / packed kernel's quad loads, ragged ends,
    single tiles and partial waves; solution against numpy."""
    rng = np.random.default_rng(n)
    main = 4.0 + rng.integers(0, 3, n).astype(np.float64)
    rows, cols, vals = [], [], []
    for i in range(n):
        for j, v in ((i - 1, -1.0), (i, main[i]), (i + 1, -1.0)):
            if 0 <= j < n:
                rows.append(i); cols.append(j); vals.append(v)
    rp = np.zeros(n + 1, np.int32)
    np.add.at(rp, np.asarray(rows) + 1, 1)
    rp = np.cumsum(rp).astype(np.int32)
    col, val = np.asarray(cols, np.int32), np.asarray(vals, np.float64)
    A = np.zeros((n, n))
    A[rows, cols] = vals
    b = rng.standard_normal(n)
    x, info = pcg_csr(rp, col, val, b, np.zeros(n), 1e-12, 500)
    assert info.converged == 1
    assert np.allclose(A @ x, b, rtol=0, atol=1e-9 * max(1.0, np.abs(b).max()))
