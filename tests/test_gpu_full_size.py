"""BASELINE.json configs at full size on one MI355X, checked through size-independent properties
(the oracle would take minutes there): symmetry (x.Ay == y.Ax), residual of the returned solution
recomputed with an independent SpMV kernel, idempotence (re-solving from the solution takes 0
iterations), rigid translations are fixed points, CSR structural invariants."""
import ctypes as C

import numpy as np
import pytest
import torch

from util import build_pyramid, feed

from adaptiveviscositysolver_amd import ViscositySolve, capi, scenes

pytestmark = pytest.mark.gpu

CONFIGS = {
    "cfg2_128_L3": lambda dev: scenes.fat_beam(128, 3, device=dev),
    "cfg3_256_L4_varvisc": lambda dev: scenes.fat_beam(256, 4, variable_viscosity=True, device=dev),
    "cfg4_512_L4": lambda dev: scenes.fat_beam(512, 4, device=dev),
}


def spmv(lib, rp, col, val, x, variant):
    y = torch.empty_like(x)
    capi.check(lib.avs_spmv_csr(len(x), rp.data_ptr(), col.data_ptr(), val.data_ptr(), x.data_ptr(), y.data_ptr(),
                                variant, 1, torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    return y


@pytest.mark.parametrize("name", list(CONFIGS))
def test_full_size_properties(name, built_lib):
    dev = torch.device("cuda:0")
    sc = CONFIGS[name](dev)
    pyr = build_pyramid(sc)
    s = ViscositySolve(sc.res, sc.dx, sc.dt, pyr.levels, device=0)
    feed(s, pyr)
    s.set_scene_fields(sc)
    ai = s.assemble()
    n, nnz = ai.n_velocity, ai.nnz
    assert n == pyr.n_velocity and nnz > 14 * n * 0.8
    rp = torch.empty(n + 1, dtype=torch.int32, device=dev)
    col = torch.empty(nnz, dtype=torch.int32, device=dev)
    val = torch.empty(nnz, dtype=torch.float64, device=dev)
    rhs = torch.empty(n, dtype=torch.float64, device=dev)
    capi.check(s.lib.avs_get_csr(s.h, rp.data_ptr(), col.data_ptr(), val.data_ptr(), rhs.data_ptr(), capi.MEM_DEVICE))
    # structural invariants: monotone row pointers, sorted unique columns, diagonal present and positive
    lens = rp[1:] - rp[:-1]
    assert int(rp[0]) == 0 and int(rp[-1]) == nnz and bool((lens > 0).all())
    same_row = torch.ones(nnz - 1, dtype=torch.bool, device=dev)
    same_row[(rp[1:-1] - 1).long()] = False
    assert bool((col[1:][same_row] > col[:-1][same_row]).all())
    rows = torch.repeat_interleave(torch.arange(n, device=dev), lens.long())
    diag = val[col.long() == rows]
    assert diag.numel() == n and bool((diag > 0).all())
    assert int(torch.bincount(lens.long()).argmax()) == 15          # uniform interior rows (cpp:539)
    # symmetry through two random vectors, with two different kernels
    g = torch.Generator(device=dev).manual_seed(11)
    x = torch.randn(n, dtype=torch.float64, device=dev, generator=g)
    y = torch.randn(n, dtype=torch.float64, device=dev, generator=g)
    ax, ay = spmv(capi.load_probe(), rp, col, val, x, 24), spmv(capi.load_probe(), rp, col, val, y, 3)
    lhs, rhs_ = float(y @ ax), float(x @ ay)
    assert abs(lhs - rhs_) <= 1e-10 * max(abs(lhs), abs(rhs_), 1.0)
    assert float(x @ ax) > 0
    # solve, then recompute the residual independently (vector kernel, reference-order matrix)
    tol = 1e-6
    info = s.solve(tol, 20000)
    assert info.converged == 1
    xs = torch.empty(n, dtype=torch.float64, device=dev)
    capi.check(s.lib.avs_get_solution(s.h, xs.data_ptr(), n, capi.MEM_DEVICE))
    r = rhs - spmv(capi.load_probe(), rp, col, val, xs, 3)
    rel = float(torch.linalg.norm(r) / torch.linalg.norm(rhs))
    assert rel <= 1.05 * tol and abs(rel - info.error) <= 0.05 * tol
    # idempotence: seam A started from the solution needs no iteration
    si = capi.SolveInfo()
    x2 = xs.clone()
    capi.check(s.lib.avs_pcg_csr(n, rp.data_ptr(), col.data_ptr(), val.data_ptr(), rhs.data_ptr(), x2.data_ptr(),
                                 1.05 * tol, 100, capi.MEM_DEVICE, 0, None, C.byref(si)))
    assert si.iterations == 0 and torch.equal(x2, xs)
    # default-tolerance solve: the reference's settings converge well inside its iteration cap
    info3 = s.solve(1e-3, 2500)
    assert info3.converged == 1 and info3.error < 1e-3
    s.close()


def test_rigid_translation_at_256(built_lib):
    dev = torch.device("cuda:0")
    sc = scenes.fat_beam(256, 4, device=dev)
    sc.velocity = scenes.constant_velocity(sc.res, (1.0, -0.5, 0.25), device=dev)
    pyr = build_pyramid(sc)
    s = ViscositySolve(sc.res, sc.dx, sc.dt, pyr.levels, device=0)
    feed(s, pyr)
    s.set_scene_fields(sc)
    s.assemble()
    info = s.solve(1e-8, 50)
    assert info.iterations == 0 and info.converged == 1
    x0 = s.initial_guess()
    tab = s.dof_table()
    want = np.array([1.0, -0.5, 0.25])[tab[:, 0] >> 8]
    assert np.array_equal(x0, want)              # restriction of a constant field is that constant, exactly
    assert np.array_equal(s.solution(), x0)


def test_distributed_assembly_at_512(built_lib):
    """The headline workload through the multi-GPU path: two virtual ranks assemble their own slabs
    (avs_dist_assemble) and solve; iteration count and solution must match the single-GPU solve."""
    import threading
    from adaptiveviscositysolver_amd import DevicePrepass
    dev = torch.device("cuda:0")
    sc = scenes.fat_beam(512, 4, device=dev)
    pp = DevicePrepass(sc.res, sc.dx, sc.levels)
    pi = pp.run(sc.liquid, sc.solid)

    def fresh():
        s = ViscositySolve(sc.res, sc.dx, sc.dt, pi.levels, device=0)
        pp.apply(s)
        s.set_scene_fields(sc)
        return s

    ref = fresh()
    ref.assemble()
    tol = 1e-6
    iref = ref.solve(tol, 20000)
    xref = torch.empty(iref.n, dtype=torch.float64, device=dev)
    capi.check(ref.lib.avs_get_solution(ref.h, xref.data_ptr(), iref.n, capi.MEM_DEVICE))
    nnz_ref = ref.info().nnz
    ref.close()
    world = 2
    lib = capi.load()
    grp = C.c_void_p()
    capi.check(lib.avs_local_group_create(world, C.byref(grp)))
    solvers = [fresh() for _ in range(world)]
    pp.close()
    out, errors = [None] * world, []

    def run(r):
        try:
            s = solvers[r]
            s.dist_init_local(grp, r)
            ai = s.dist_assemble()
            info = s.dist_solve(tol, 20000)
            x = torch.empty(iref.n, dtype=torch.float64, device=dev)
            capi.check(lib.avs_dist_get_solution(s.h, x.data_ptr(), iref.n, capi.MEM_DEVICE))
            out[r] = (info.iterations, info.converged, ai.nnz, s.plan_sizes.n_own, s.matrix_format().bytes_per_nonzero, x)
        except Exception as e:  # pragma: no cover
            errors.append((r, e))

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=600)
    assert not errors, errors
    assert sum(o[2] for o in out) == nnz_ref and sum(o[3] for o in out) == iref.n
    for it, conv, _, n_own, bpn, x in out:
        # ~1740 iterations to 1e-6: the count moves by a few with the summation order of the dot products (the single-GPU and
        # the distributed loop fold their partial sums differently), so the check is 0.5 %, not equality
        assert conv == 1 and abs(it - iref.iterations) <= max(3, iref.iterations // 200)
        assert 0.4 * iref.n < n_own < 0.6 * iref.n          # slabs balanced by raw triplet counts
        assert bpn == 4                                      # each rank's own dictionary, packed form
        assert float(torch.linalg.norm(x - xref) / torch.linalg.norm(xref)) < 1e-5
    for s in solvers:
        s.close()
    lib.avs_local_group_destroy(grp)


@pytest.mark.skipif(__import__("os").environ.get("AVS_SKIP_SLOW") == "1", reason="AVS_SKIP_SLOW=1")
def test_headline_512_against_the_oracle(built_lib):
    """The BASELINE headline workload (512^3, 4 levels) compared with the CPU oracle ITSELF, once: DOF counts, the CSR
    (bit-exact pattern, values and rhs), the warm start, the iteration count at the reference's tolerance 1e-3 and the
    solution at 1e-8.  Slow (the oracle needs 2-4 minutes on the host cores); everything else at this size is
    property-checked (test_full_size_properties)."""
    import os
    import time
    from util import oracle_for_scene
    dev = torch.device("cuda:0")
    t0 = time.time()
    sc_h = scenes.fat_beam(512, 4)
    sc = scenes.to_device(sc_h, dev)      # identical inputs on both sides
    pyr = build_pyramid(sc)
    s = ViscositySolve(sc.res, sc.dx, sc.dt, pyr.levels, device=0)
    feed(s, pyr)
    s.set_scene_fields(sc)
    ai = s.assemble()
    fmt0 = s.matrix_format()
    # the comparison below must cover the kernel the headline runs: the brick-structured form, chosen by the (deterministic) default mode
    assert fmt0.brick_tiles > 0 and fmt0.brick_pattern_rows > 0.9 * ai.n_velocity and fmt0.brick_walk == 0, (fmt0.brick_tiles, fmt0.brick_walk)
    sys_path = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import sys
    sys.path.insert(0, sys_path)
    from bench import cpu_quota
    quota = cpu_quota()
    threads = max(1, min(len(os.sched_getaffinity(0)) // 2, 64 if quota is None else int(quota)))
    say = lambda *a: print(f"[{time.time() - t0:6.0f} s]", *a, flush=True)
    say("gpu side assembled; oracle threads", threads, "quota", quota)
    o = oracle_for_scene(sc_h)
    o.prepass()
    assert (pyr.levels, pyr.n_velocity, pyr.n_edge, pyr.n_center) == (o.levels, o.count(0), o.count(1), o.count(2))
    say("oracle pre-pass done")
    o.hot_path()
    t1 = time.time()
    say("oracle hot path done")
    A = o.csr()
    rp, col, val, rhs = s.csr()
    assert ai.nnz == len(A.col) and ai.raw_triplets == o.raw_triplets
    assert np.array_equal(rp, A.row_ptr.astype(np.int32)) and np.array_equal(col, A.col)
    assert np.array_equal(val, A.val) and np.array_equal(rhs, A.rhs)
    assert np.array_equal(s.initial_guess(), o.initial_guess())
    del rp, col, val, rhs
    # reference settings (tol 1e-3): the iteration count is the bench's "cg_iterations_per_step"
    i3 = s.solve(1e-3, 2500)
    x3 = s.solution()
    say("CSR / rhs / x0 bit-exact; oracle solve 1e-3 ...")
    xo3, io3 = o.solve(1e-3, 2500, threads=threads)
    say("oracle 1e-3 done", io3.iterations, "iterations,", io3.seconds, "s")
    # the count moves by a few with the summation order of the dot products (device: fixed trees; oracle: per-thread shares added in
    # thread order -- deterministic for a given thread count since round 2, 1271 vs 1275 on 16 threads): 0.5 %, as at 1e-8 below
    assert i3.converged == 1 and abs(i3.iterations - io3.iterations) <= max(3, io3.iterations // 200), (i3.iterations, io3.iterations)
    assert float(np.linalg.norm(x3 - xo3) / np.linalg.norm(xo3)) < 1e-5
    # ... and against the oracle with SERIAL dot products (row-parallel SpMV only: what Eigen threads; independent of the thread count).
    # Three summation orders of the same dot products (serial, 16 shares, the device's trees) give counts a few apart: the spread between
    # the two ORACLE runs is the yardstick for the device's distance to either (round-3 review, weak #2)
    from oracle import oracle as _O
    xs3, is3 = _O.pcg_csr_ex(A.row_ptr, A.col, A.val, A.rhs, o.initial_guess(), 1e-3, 2500, spmv_threads=threads, vec_threads=1)
    say("oracle 1e-3 with serial dots:", is3.iterations, "iterations")
    spread = abs(is3.iterations - io3.iterations)
    assert abs(i3.iterations - is3.iterations) <= max(3, 2 * spread, is3.iterations // 200), (i3.iterations, is3.iterations, io3.iterations)
    assert float(np.linalg.norm(x3 - xs3) / np.linalg.norm(xs3)) < 1e-5
    print(f"512^3 iteration counts at 1e-3: device {i3.iterations}, oracle serial dots {is3.iterations}, oracle {threads}-thread shares {io3.iterations}")
    # tight tolerance: the velocity field itself (north_star: 1e-5 relative L2)
    i8 = s.solve(1e-8, 20000)
    x8 = s.solution()
    xo8, io8 = o.solve(1e-8, 20000, threads=threads)
    assert i8.converged == 1 and abs(i8.iterations - io8.iterations) <= max(3, io8.iterations // 200), (i8.iterations, io8.iterations)
    rel = float(np.linalg.norm(x8 - xo8) / np.linalg.norm(xo8))
    assert rel < 1e-5
    print(f"512^3 vs oracle: iterations {i3.iterations}/{io3.iterations} (1e-3), {i8.iterations}/{io8.iterations} (1e-8), rel L2 {rel:.2e}; "
          f"oracle pre-pass+assembly {t1 - t0:.0f} s, total {time.time() - t0:.0f} s on {threads} threads")
    s.close()


def test_context_reuse_with_another_scene_size(built_lib):
    """One avs_ctx across frames whose DOF count changes: the PCG work space follows the system size."""
    dev = torch.device("cuda:0")
    s = None
    counts = []
    for scene in (scenes.sphere(64, 3, radius=0.3, device=dev), scenes.sphere(64, 3, radius=0.2, device=dev),
                  scenes.sphere(64, 3, radius=0.35, device=dev)):
        pyr = build_pyramid(scene)
        if s is None:
            s = ViscositySolve(scene.res, scene.dx, scene.dt, pyr.levels, device=0)
        assert pyr.levels == s.levels
        feed(s, pyr)
        s.set_scene_fields(scene)
        s.assemble()
        info = s.solve(1e-8, 5000)
        assert info.converged == 1 and info.n == pyr.n_velocity
        counts.append(info.n)
        rp, col, val, rhs = s.csr()
        import scipy.sparse as sp
        A = sp.csr_matrix((val, col, rp.astype(np.int64)), shape=(len(rhs), len(rhs)))
        x = s.solution()
        assert np.linalg.norm(A @ x - rhs) <= 2e-8 * np.linalg.norm(rhs)
    assert len(set(counts)) == 3


@pytest.mark.parametrize("thickness,want_levels", [(32, 4), (64, 5)])
def test_config5_thin_sheet_1024_properties(thickness, want_levels, built_lib):
    """BASELINE configs[4]: 1024^3-equivalent, 5 requested levels, thin free-surface sheet, on one GPU end to end through the HIP pre-pass:
    structural invariants, symmetry through two kernels, independent residual of the solve, default-tolerance convergence (the oracle
    would need tens of minutes).  Half-thickness 16 dx (SURVEY 8(d) Config 5): 18.3 M rows / 270 M non-zeros, FOUR levels appear
    (oct.cpp:198-211 caps them by the sheet's thickness); half-thickness 32 dx: all FIVE levels really appear at 1024^3."""
    from adaptiveviscositysolver_amd import DevicePrepass
    dev = torch.device("cuda:0")
    sc = scenes.thin_sheet(1024, 5, thickness_cells=thickness, device=dev)
    pp = DevicePrepass(sc.res, sc.dx, sc.levels)
    pi = pp.run(sc.liquid, sc.solid)
    assert pi.levels == want_levels
    s = ViscositySolve(sc.res, sc.dx, sc.dt, pi.levels, device=0, probe=True)   # (bench_spmv below: a probe-library entry)
    pp.apply(s)
    s.set_scene_fields(sc)
    pp.close()
    ai = s.assemble()
    n, nnz = ai.n_velocity, ai.nnz
    assert n == pi.n_velocity and n > 10_000_000 and nnz > 14 * n * 0.8
    rp = torch.empty(n + 1, dtype=torch.int32, device=dev)
    col = torch.empty(nnz, dtype=torch.int32, device=dev)
    val = torch.empty(nnz, dtype=torch.float64, device=dev)
    rhs = torch.empty(n, dtype=torch.float64, device=dev)
    capi.check(s.lib.avs_get_csr(s.h, rp.data_ptr(), col.data_ptr(), val.data_ptr(), rhs.data_ptr(), capi.MEM_DEVICE))
    lens = rp[1:] - rp[:-1]
    assert int(rp[0]) == 0 and int(rp[-1]) == nnz and bool((lens > 0).all())
    assert int(torch.bincount(lens.long()).argmax()) == 15
    g = torch.Generator(device=dev).manual_seed(5)
    x = torch.randn(n, dtype=torch.float64, device=dev, generator=g)
    y = torch.randn(n, dtype=torch.float64, device=dev, generator=g)
    ax, ay = spmv(capi.load_probe(), rp, col, val, x, 24), spmv(capi.load_probe(), rp, col, val, y, 3)
    lhs, rhs_ = float(y @ ax), float(x @ ay)
    assert abs(lhs - rhs_) <= 1e-10 * max(abs(lhs), abs(rhs_), 1.0) and float(x @ ax) > 0
    del x, y, ax, ay
    s.bench_spmv(0, 2)                                       # compressed form == plain CSR bit for bit at this size too
    info = s.solve(1e-3, 2500)
    assert info.converged == 1 and info.error < 1e-3
    tol = 1e-6
    info6 = s.solve(tol, 20000)
    assert info6.converged == 1
    xs = torch.empty(n, dtype=torch.float64, device=dev)
    capi.check(s.lib.avs_get_solution(s.h, xs.data_ptr(), n, capi.MEM_DEVICE))
    r = rhs - spmv(capi.load_probe(), rp, col, val, xs, 3)
    rel = float(torch.linalg.norm(r) / torch.linalg.norm(rhs))
    assert rel <= 1.05 * tol and abs(rel - info6.error) <= 0.05 * tol
    fmt = s.matrix_format()
    print(f"config 5 (half-thickness {thickness // 2} dx): levels {pi.levels}, n {n}, nnz {nnz}, {fmt.bytes_per_nonzero} B/nnz, iterations {info.iterations} (1e-3) / {info6.iterations} (1e-6)")
    s.close()
