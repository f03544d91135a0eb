"""CPU tests of the oracle: maths-derived known answers and the reference's debug invariants.

The reference ships no vectors, so the oracle is pinned by
  * the uniform-interior row of SURVEY.md A.8 (derived from cpp:1827, 1934, 2155, 2284, 2439-2451, 2768),
  * symmetry / positive definiteness of A (cpp:424),
  * rigid translations being fixed points,
  * an independent solve of the same system with scipy,
  * the invariants of HDK_OctreeGrid::unitTest (oct.cpp:984-1275) and of the stress/velocity
    debug tests (cpp:2896-3298).
"""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla
import torch

from adaptiveviscositysolver_amd import scenes
from oracle import oracle as O
from util import oracle_for_scene, rel_l2


def full_domain_scene(n=16, mu=8.0, rho=4.0, dt=0.5):
    res = (n, n, n)
    liquid = torch.full((n, n, n), -100.0, dtype=torch.float32)
    vel = scenes.smooth_velocity(res, 1.0 / n)
    return scenes.Scene(res=res, dx=1.0 / n, dt=dt, levels=1, liquid=liquid, viscosity=mu, density=rho, velocity=vel)


def test_known_answer_row_uniform_interior():
    """A_ii = rho + 8k, 2 x -2k, 4 x -k, 8 x +-k with k = dt*mu/dx^2, exactly."""
    sc = full_domain_scene()
    o = oracle_for_scene(sc)
    o.prepass()
    assert o.levels == 1
    o.hot_path()
    A = o.csr()
    n = sc.res[0]
    kappa = sc.dt * sc.viscosity / sc.dx ** 2
    rho = sc.density
    vidx = [o.index(O.I_VELOCITY, 0, a) for a in range(3)]
    x0 = o.initial_guess()
    M = A.to_scipy()
    for axis in range(3):
        f = [n // 2, n // 2, n // 2]
        row = int(vidx[axis][f[2], f[1], f[0]])
        assert row >= 0
        lo, hi = A.row_ptr[row], A.row_ptr[row + 1]
        cols, vals = A.col[lo:hi], A.val[lo:hi]
        assert len(cols) == 15
        d = dict(zip(cols.tolist(), vals.tolist()))
        assert d[row] == rho + 8 * kappa
        def fid(ax, off):
            g = [f[0] + off[0], f[1] + off[1], f[2] + off[2]]
            return int(vidx[ax][g[2], g[1], g[0]])
        e = [np.eye(3, dtype=int)[a] for a in range(3)]
        for s in (-1, 1):
            assert d[fid(axis, s * e[axis])] == -2 * kappa
            for b in range(3):
                if b != axis:
                    assert d[fid(axis, s * e[b])] == -kappa
        others = sorted(v for c, v in d.items() if c != row and abs(v) == kappa and
                        not any(c == fid(axis, s * e[b]) for s in (-1, 1) for b in range(3) if b != axis))
        assert others == [-kappa] * 4 + [kappa] * 4
        assert abs(vals.sum() - rho) < 1e-9 * kappa          # row sum = rho
        assert A.rhs[row] == rho * x0[row]                    # rhs_i = rho u_i^n
    assert abs(M - M.T).max() == 0.0                          # all operands are binary fractions here


@pytest.mark.parametrize("maker", [
    lambda: scenes.fat_beam(32, 3),
    lambda: scenes.fat_beam(32, 3, wall=True),
    lambda: scenes.fat_beam(32, 4, variable_viscosity=True),
    lambda: scenes.sphere(32, 3),
])
def test_matrix_is_symmetric_positive_definite(maker):
    """A = M_u + sum_s w_s d_s d_s^T (cpp:424): symmetric, w_s >= 0, M_u >= 0 => positive SEMI-definite.
    (Fringe faces may be active with zero face weight -- zero mass -- so A can be singular; the
    right-hand side is then consistent and CG stays in the range space.)"""
    o = oracle_for_scene(maker())
    o.prepass()
    o.hot_path()
    A = o.csr().to_scipy()
    asym = abs(A - A.T).max()
    assert asym <= 1e-12 * abs(A).max()
    assert (A.diagonal() > 0).all()
    assert (o.edge_stencils()["weight"] >= 0).all() and (o.center_stencils()["weight"] >= 0).all()
    rng = np.random.default_rng(3)
    for _ in range(16):
        v = rng.standard_normal(A.shape[0])
        assert v @ (A @ v) >= -1e-9 * abs(A).max() * (v @ v)


def test_rows_have_15_entries_in_uniform_regions_and_selves_found():
    o = oracle_for_scene(scenes.fat_beam(32, 3))
    o.prepass()
    o.hot_path()                       # raises if a row's stencil lacks the row DOF (assert foundSelf cpp:2436)
    rl = np.diff(o.csr().row_ptr)
    assert np.bincount(rl).argmax() == 15          # reserve factor of cpp:539, 546


def test_rigid_translation_is_fixed_point():
    sc = scenes.sphere(32, 3)
    sc.velocity = scenes.constant_velocity(sc.res, (0.5, -2.0, 1.25))
    o = oracle_for_scene(sc)
    o.prepass()
    o.hot_path()
    A = o.csr()
    u = o.initial_guess()
    r = A.rhs - A.to_scipy() @ u
    assert np.linalg.norm(r) <= 1e-12 * np.linalg.norm(A.rhs)
    x, info = o.solve(1e-8, 100)
    assert info.iterations == 0 and np.array_equal(x, u)


def test_pcg_matches_independent_scipy_cg():
    o = oracle_for_scene(scenes.fat_beam(32, 3, variable_viscosity=True))
    o.prepass()
    o.hot_path()
    A = o.csr()
    M = A.to_scipy()
    x0 = o.initial_guess()
    x, info = o.solve(1e-12, 20000)
    dinv = sp.diags(1.0 / M.diagonal())
    xs, flag = spla.cg(M, A.rhs, x0=x0, rtol=1e-13, atol=0.0, maxiter=20000, M=dinv)
    assert flag == 0
    assert rel_l2(x, xs) < 1e-8
    assert info.error <= 1e-12
    assert np.linalg.norm(A.rhs - M @ x) <= 1e-11 * np.linalg.norm(A.rhs)
    # the restated Eigen loop counts iterations like Eigen: tolerance 1e-3 takes fewer steps and
    # stops at the first |r|^2 < tol^2 |b|^2
    x3, info3 = o.solve(1e-3, 2500)
    assert 0 < info3.iterations < info.iterations and info3.error < 1e-3
    # parallel variant gives the same answer up to summation order
    xp, infop = o.solve(1e-12, 20000, threads=4)
    assert rel_l2(xp, x) < 1e-9
    # ... and is deterministic for a given thread count: the dot products add per-thread shares in thread order (an OpenMP
    # `reduction` combined them in arrival order, and the iteration count of a long solve wandered from run to run)
    for _ in range(3):
        xq, infoq = o.solve(1e-12, 20000, threads=4)
        assert infoq.iterations == infop.iterations and np.array_equal(xq, xp)


def test_pcg_zero_rhs_and_converged_guess():
    A = sp.diags([[4.0] * 10, [-1.0] * 9, [-1.0] * 9], [0, 1, -1]).tocsr()
    x, info = O.pcg_csr(A.indptr, A.indices, A.data, np.zeros(10), np.ones(10), 1e-3, 10)
    assert info.iterations == 0 and not x.any() and info.error == 0     # x.setZero() branch
    b = A @ np.arange(10.0)
    x, info = O.pcg_csr(A.indptr, A.indices, A.data, b, np.arange(10.0), 1e-3, 10)
    assert info.iterations == 0 and np.array_equal(x, np.arange(10.0))  # residual below threshold at entry


def test_enhanced_gradients_only_change_transition_rows():
    sc = scenes.sphere(32, 3)
    on = oracle_for_scene(sc, enhanced=True)
    off = oracle_for_scene(sc, enhanced=False)
    for o in (on, off):
        o.prepass()
        o.hot_path()
    A, B = on.csr(), off.csr()
    assert A.n == B.n
    D = (A.to_scipy() - B.to_scipy()).tocsr()
    D.eliminate_zeros()
    changed = int((np.diff(D.indptr) > 0).sum())
    assert 0 < changed < 0.5 * A.n
    # a 1-level tree has no transitions at all: the flag must not matter
    sc1 = full_domain_scene()
    p, q = oracle_for_scene(sc1, enhanced=True), oracle_for_scene(sc1, enhanced=False)
    for o in (p, q):
        o.prepass()
        o.hot_path()
    assert np.array_equal(p.csr().val, q.csr().val) and np.array_equal(p.csr().col, q.csr().col)


# ---- reference invariants --------------------------------------------------------------------
def _ancestor(lab_hi, shift, shape):
    idx = np.indices(shape) >> shift
    return lab_hi[idx[0], idx[1], idx[2]]


@pytest.mark.parametrize("maker", [lambda: scenes.fat_beam(32, 4), lambda: scenes.sphere(64, 4),
                                   lambda: scenes.fat_beam(64, 3, wall=True)])
def test_octree_invariants(maker):
    """activeCountUnitTest / upAdjacentUnitTest / activeUnitTest, oct.cpp:984-1275."""
    o = oracle_for_scene(maker())
    o.prepass()
    L = o.levels
    labs = [o.labels(l) for l in range(L)]
    shape = labs[0].shape
    col = np.stack([_ancestor(labs[l], l, shape) for l in range(L)])      # [level, z, y, x]
    n_active = (col == O.ACTIVE).sum(axis=0)
    base = col[0]
    assert ((n_active == 1) | (base == O.INACTIVE)).all()                  # one leaf per column
    assert (n_active[base == O.INACTIVE] == 0).all()
    for l in range(1, L):
        anc = col[l]
        assert np.isin(anc[base == O.INACTIVE], (O.INACTIVE, O.DOWN)).all()
        assert (anc[base == O.ACTIVE] == O.DOWN).all()
    # below the leaf everything is UP, above it everything is DOWN
    leaf_level = (col == O.ACTIVE).argmax(axis=0)
    for l in range(L):
        inside = base != O.INACTIVE
        assert (col[l][inside & (leaf_level > l)] == O.UP).all()
        assert (col[l][inside & (leaf_level < l)] == O.DOWN).all()
    # 2:1 face grading: face-adjacent leaves differ by at most one level
    for d in range(3):
        a = [slice(None)] * 3
        b = [slice(None)] * 3
        a[d], b[d] = slice(0, -1), slice(1, None)
        both = (base[tuple(a)] != O.INACTIVE) & (base[tuple(b)] != O.INACTIVE)
        assert (np.abs(leaf_level[tuple(a)] - leaf_level[tuple(b)])[both] <= 1).all()


def test_velocity_and_stress_label_invariants():
    """octreeVelocityUnitTest / edgeStressUnitTest / centerStresUnitTest essentials, cpp:2896-3298."""
    o = oracle_for_scene(scenes.sphere(64, 4))
    o.prepass()
    for l in range(o.levels):
        lab = o.labels(l)
        for axis in range(3):
            v = o.index(O.I_VELOCITY, l, axis)
            d = 2 - axis
            if l > 0:
                assert not np.isin(v, (O.SOLIDBOUNDARY, O.OUTSIDE)).any()   # only at level 0
            a = [slice(None)] * 3
            b = [slice(None)] * 3
            a[d], b[d] = slice(0, -1), slice(1, None)
            inner = [slice(None)] * 3
            inner[d] = slice(1, -1)
            act = v[tuple(inner)] >= 0
            la, lb = lab[tuple(a)][act], lab[tuple(b)][act]
            ok = ((la == O.ACTIVE) & (lb == O.ACTIVE)) | ((la == O.ACTIVE) & (lb == O.UP)) | ((la == O.UP) & (lb == O.ACTIVE))
            assert ok.all()
        c = o.index(O.I_CENTER, l)
        assert (lab[c >= 0] == O.ACTIVE).all()
    # ids are a permutation-free numbering 0..n-1 in (level, axis) blocks
    for kind in (O.I_VELOCITY, O.I_EDGE):
        seen = []
        for l in range(o.levels):
            for axis in range(3):
                g = o.index(kind, l, axis)
                ids = np.sort(g[g >= 0])
                if len(ids):
                    seen.append((ids[0], ids[-1], len(ids)))
        start = 0
        for lo, hi, cnt in seen:
            assert lo == start and hi == start + cnt - 1
            start += cnt
        assert start == o.count(kind)


# ---- post-solve transfer (cpp:655-707) ---------------------------------------------------------
def test_transfer_reproduces_rigid_translation_exactly():
    """Interpolation weights sum to one: a constant field survives setOctreeVelocity -> node values ->
    interpSPGrid -> regular grid bit for bit (no solid)."""
    sc = scenes.fat_beam(64, 4)
    cv = (0.5, -2.0, 1.25)
    sc.velocity = scenes.constant_velocity(sc.res, cv)
    o = oracle_for_scene(sc)
    o.prepass()
    o.build_regular_indices()
    o.hot_path()
    x, info = o.solve(1e-8, 50)
    assert info.iterations == 0
    out = o.transfer_to_regular_grid(x)
    for a in range(3):
        assert np.array_equal(out[a], np.full_like(out[a], cv[a]))


def test_transfer_touches_only_regular_dofs_and_copies_level0_faces():
    sc = scenes.sphere(32, 3)
    o = oracle_for_scene(sc)
    o.prepass()
    o.build_regular_indices()
    o.hot_path()
    x, _ = o.solve(1e-10, 5000)
    out = o.transfer_to_regular_grid(x)
    total = 0
    for a in range(3):
        ri, oi = o.regular_index(a), o.index(O.I_VELOCITY, 0, a)
        vin = sc.velocity[a].numpy()
        assert np.array_equal(out[a][ri == O.UNASSIGNED], vin[ri == O.UNASSIGNED])       # untouched faces
        direct = (ri >= 0) & (oi >= 0)
        assert np.array_equal(out[a][direct], x[oi[direct]].astype(np.float32))           # cpp:2856-2857
        interp = (ri >= 0) & (oi == O.UNASSIGNED)
        assert interp.sum() > 0 and np.isfinite(out[a][interp]).all()
        lo, hi = x.min(), x.max()
        assert out[a][interp].min() >= lo - 1e-6 and out[a][interp].max() <= hi + 1e-6   # convex-ish combination + bubble
        total += int((ri >= 0).sum())
    assert total == o.regular_count
    # node labels: every node is inactive or active after distributeNodeValuesDown (no DEPENDENT left)
    for l in range(o.levels):
        lab, _ = o.node_grid(l)
        assert set(np.unique(lab)) <= {0, 1}


def leaving_box_scene():
    """Liquid that leaves the domain through its top border while the octree's top level is still coarsening there:
    the reference indexes past its level arrays in getEdgeStressFaces (cpp:1853 / cpp:1888 with level + 1 ==
    octreeLevels).  Found by tools/stress_parity.py (seed 7, case 21)."""
    from adaptiveviscositysolver_amd import scenes
    res, n = (64, 32, 64), 64
    dx = 1.0 / n
    liquid = scenes.box_sdf(res, dx, (0.4644382608195187, 0.2995961010117988, 0.6258390038916162),
                            (0.18445403858748277, 0.17004666199877277, 0.3982372146871515))
    solid = scenes.wall_sdf(res, dx, 0.2644382608195187)
    return scenes.Scene(res=res, dx=dx, dt=0.02, levels=4, liquid=liquid, solid=solid, viscosity=1000.0, density=1000.0,
                        velocity=scenes.smooth_velocity(res, dx, gravity_dt=0.1), name="leaving_box")


def test_state_the_reference_asserts_on_is_rejected_not_crashed():
    from util import oracle_for_scene
    o = oracle_for_scene(leaving_box_scene())
    o.prepass()
    with pytest.raises(RuntimeError):
        o.build_stencils()


def _eigen_redux_f32(a, b):
    """Eigen 3.3 / 3.4, Core/Redux.h, redux_impl<..., LinearVectorizedTraversal, NoUnrolling> for a.cwiseProduct(b).sum() of two VectorXf
    with 4-float packets (SSE2: the reference's CMakeLists.txt sets no -march): two packet accumulators on alternate packets, added lane
    by lane, a last odd packet, predux as (l0 + l2) + (l1 + l3), then the scalar tail.  Every operation rounds to float."""
    f = np.float32
    n = len(a)
    prod = (a.astype(f) * b.astype(f)).astype(f)
    size1, size2 = (n // 4) * 4, (n // 8) * 8
    if size1 == 0:
        res = f(0) if n == 0 else prod[0]
        for i in range(1, n):
            res = f(res + prod[i])
        return res
    p0 = prod[0:4].copy()
    if size1 > 4:
        p1 = prod[4:8].copy()
        for i in range(8, size2, 8):
            p0 = (p0 + prod[i:i + 4]).astype(f)
            p1 = (p1 + prod[i + 4:i + 8]).astype(f)
        p0 = (p0 + p1).astype(f)
        if size1 > size2:
            p0 = (p0 + prod[size2:size2 + 4]).astype(f)
    res = f(f(p0[0] + p0[2]) + f(p0[1] + p0[3]))
    for i in range(size1, n):
        res = f(res + prod[i])
    return res


@pytest.mark.parametrize("n", [0, 1, 3, 4, 5, 7, 8, 9, 12, 13, 16, 31, 1000, 4099])
def test_float_cg_dots_follow_eigens_redux_order(n):
    """The oracle's float CG (SolveType = fpreal32) folds its dot products as Eigen's own float reduction does (oracle/avs_oracle.c,
    dot_f32_eigen): bit for bit against a numpy restatement of Redux.h, and different from the left-to-right sum on long vectors."""
    import ctypes as C
    from oracle.oracle import lib
    L = lib()
    L.orc_dot_f32.restype = C.c_float
    L.orc_dot_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    rng = np.random.default_rng(n + 1)
    a = (rng.standard_normal(n) * 10.0 ** rng.integers(-2, 3, n)).astype(np.float32)
    b = rng.standard_normal(n).astype(np.float32)
    got = np.float32(L.orc_dot_f32(a.ctypes.data, b.ctypes.data, n))
    want = _eigen_redux_f32(a, b)
    assert got.view(np.int32) == np.float32(want).view(np.int32), (n, got, want)
    if n >= 1000:
        serial = np.float32(0)
        for v in (a * b).astype(np.float32):
            serial = np.float32(serial + v)
        assert abs(float(got) - float(np.dot(a.astype(np.float64), b.astype(np.float64)))) <= abs(float(serial) - float(np.dot(a.astype(np.float64), b.astype(np.float64)))) + 1e-3
