"""Slab-local pre-pass and slab-local assembly (round 6; SURVEY 8(e): "each GPU assembles the rows it owns from its slab of the
pyramids plus a halo").  Virtual ranks (one host thread, one pre-pass object and one solve context each, one GPU):

  * the pre-pass of a rank's window -- labels, index lattices (GLOBAL ids), weights -- equals the corresponding part of the
    single-rank pre-pass bit for bit, the DOF counts and the levels cap are the global ones on every rank;
  * the slab-local assembly gives, for the same cuts, the same local systems as the replicated-index avs_dist_assemble: row
    pointers, local column ids, send lists, peers, counts, and a partitioned solve whose solution has the same BITS;
  * the rank's work scales with its window: the DOFs a rank sweeps are the window's, not the octree's.
"""
import ctypes as C
import threading

import numpy as np
import pytest
import torch

from adaptiveviscositysolver_amd import DevicePrepass, ViscositySolve, capi, scenes

pytestmark = pytest.mark.gpu

SCENES = {
    "beam64_L3": lambda dev: scenes.fat_beam(64, 3, device=dev),
    "sphere64_L4": lambda dev: scenes.sphere(64, 4, device=dev),
    "tank64_L3": lambda dev: scenes.tank(64, 3, device=dev),
    "noncubic": lambda dev: scenes.fat_beam(64, 3, res=(64, 32, 32), device=dev),
    "varvisc128_L4": lambda dev: scenes.fat_beam(128, 4, variable_viscosity=True, device=dev),
    "sheet128_L4": lambda dev: scenes.thin_sheet(128, 4, thickness_cells=12, device=dev),
    "levels_capped": lambda dev: scenes.fat_beam(16, 6, device=dev),
}


def _uniform_cuts(extent, world, step=4):
    cuts = [0]
    for r in range(1, world):
        cuts.append(min(extent, (extent * r // world) // step * step))
    cuts.append(extent)
    return np.asarray(cuts, np.int32)


def _run_threads(world, fn):
    results, errors = [None] * world, []

    def run(r):
        try:
            results[r] = fn(r)
        except Exception as e:  # pragma: no cover
            import traceback
            errors.append((r, e, traceback.format_exc()))

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=600)
    assert not errors, errors
    assert all(r is not None for r in results)
    return results


def _axis_slice(arr, axis, lo, hi):
    """arr is (z, y, x): entries [lo, hi) along the lattice axis `axis` (0 = x)."""
    sl = [slice(None)] * 3
    sl[2 - axis] = slice(lo, hi)
    return arr[tuple(sl)]


def _compare_window(pp, ref, levels, axis, lo, hi, res):
    """every lattice of the rank's pre-pass `pp` against the single-rank arrays `ref` inside the window"""
    for l in range(levels):
        n_l = res[axis] >> l
        to_end = hi[l] >= n_l
        a, b = int(lo[l]), int(hi[l])
        lab = pp.labels(l)
        assert np.array_equal(_axis_slice(lab, axis, max(a - 1, 0), min(b + 1, n_l)), _axis_slice(ref["labels"][l], axis, max(a - 1, 0), min(b + 1, n_l))), ("labels", l)
        for ax in range(3):
            for kind, key in ((capi.INDEX_VELOCITY, "vidx"), (capi.INDEX_EDGE, "eidx")):
                full = ref[key][l][ax]
                ext = full.shape[2 - axis]
                e = ext if to_end else b
                got = pp.index(kind, l, ax)
                assert np.array_equal(_axis_slice(got, axis, a, e), _axis_slice(full, axis, a, e)), (key, l, ax)
        got = pp.index(capi.INDEX_CENTER, l)
        assert np.array_equal(_axis_slice(got, axis, a, b), _axis_slice(ref["cidx"][l], axis, a, b)), ("cidx", l)
    a, b = int(lo[0]), int(hi[0])
    n0 = res[axis]
    for kind, key in ((capi.FIELD_CENTER_WEIGHTS, "cw"), (capi.FIELD_EDGE_WEIGHTS, "ew"), (capi.FIELD_FACE_WEIGHTS, "fw")):
        for ax in range(3 if key != "cw" else 1):
            full = ref[key][ax]
            ext = full.shape[2 - axis]
            e = ext if b >= n0 else b
            got = pp.weights(kind, ax)
            assert np.array_equal(_axis_slice(got, axis, a, e), _axis_slice(full, axis, a, e)), (key, ax)
    for ax in range(3):
        full = ref["ridx"][ax]
        e = full.shape[2 - axis] if b >= n0 else b
        assert np.array_equal(_axis_slice(pp.regular_index(ax), axis, a, e), _axis_slice(full, axis, a, e)), ("ridx", ax)


def _reference_arrays(sc):
    pp = DevicePrepass(sc.res, sc.dx, sc.levels)
    info = pp.run(sc.liquid, sc.solid)
    L = info.levels
    ref = {"labels": [pp.labels(l) for l in range(L)],
           "vidx": [[pp.index(capi.INDEX_VELOCITY, l, a) for a in range(3)] for l in range(L)],
           "eidx": [[pp.index(capi.INDEX_EDGE, l, a) for a in range(3)] for l in range(L)],
           "cidx": [pp.index(capi.INDEX_CENTER, l) for l in range(L)],
           "cw": [pp.weights(capi.FIELD_CENTER_WEIGHTS)],
           "ew": [pp.weights(capi.FIELD_EDGE_WEIGHTS, a) for a in range(3)],
           "fw": [pp.weights(capi.FIELD_FACE_WEIGHTS, a) for a in range(3)],
           "ridx": [pp.regular_index(a) for a in range(3)]}
    counts = (info.levels, info.n_velocity, info.n_edge, info.n_center, info.n_regular)
    pp.close()
    return ref, counts


@pytest.mark.parametrize("scene,world,axis", [("beam64_L3", 2, 0), ("beam64_L3", 4, 0), ("beam64_L3", 3, 1), ("sphere64_L4", 2, 2),
                                              ("sphere64_L4", 4, 0), ("tank64_L3", 3, 0), ("tank64_L3", 2, 2), ("noncubic", 4, 0),
                                              ("varvisc128_L4", 8, 0), ("sheet128_L4", 4, 1), ("sheet128_L4", 3, 2), ("levels_capped", 2, 0)])
def test_window_of_the_prepass_equals_the_single_rank_prepass(scene, world, axis, built_lib):
    """The hosted route: the all-reduce is the CALLER's (here: host threads, hipMemcpy through ctypes)."""
    dev = torch.device("cuda:0")
    sc = scenes.to_device(SCENES[scene]("cpu"), dev)
    ref, counts = _reference_arrays(sc)
    cuts = _uniform_cuts(sc.res[axis], world)
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    barrier = threading.Barrier(world)
    shared = {}
    lock = threading.Lock()

    def make_allreduce(rank):
        def allreduce(ptr, count, stream):
            assert hip.hipStreamSynchronize(C.c_void_p(stream)) == 0
            mine = np.empty(count, np.int32)
            assert hip.hipMemcpy(mine.ctypes.data, C.c_void_p(ptr), count * 4, 2) == 0   # device to host
            barrier.wait()
            if rank == 0:
                shared["sum"] = np.zeros(count, np.int64)
            barrier.wait()
            with lock:
                shared["sum"] += mine
            barrier.wait()
            tot = shared["sum"].astype(np.int32)
            assert hip.hipMemcpy(C.c_void_p(ptr), tot.ctypes.data, count * 4, 1) == 0    # host to device
            barrier.wait()
        return allreduce

    def rank_fn(r):
        pp = DevicePrepass(sc.res, sc.dx, sc.levels)
        pp.set_slab(axis, cuts, r, make_allreduce(r))
        out = []
        for frame in range(3):   # frames 2 and 3 run on the temporal-reuse records of the two allocations
            info = pp.run(sc.liquid, sc.solid)
            assert (info.levels, info.n_velocity, info.n_edge, info.n_center, info.n_regular) == counts, (r, frame)
            if info.levels == 0:
                continue
            lo, hi, nw = pp.window()
            _compare_window(pp, ref, info.levels, axis, lo, hi, sc.res)
            out.append((lo.copy(), hi.copy(), nw))
        pp.close()
        return out or True

    res = _run_threads(world, rank_fn)
    if counts[0] == 0:
        return
    # the windows cover the lattice; a rank's window holds a part of the DOFs (not all of them, once the slabs are thin enough)
    nw_sum = sum(r[-1][2][0] for r in res)
    assert nw_sum >= counts[1]
    if world >= 4 and sc.res[axis] >= 64:
        assert max(r[-1][2][0] for r in res) < counts[1]


def test_window_follows_moving_cuts_and_a_moving_liquid(built_lib):
    """Frame after frame on ONE pre-pass object per rank: the cuts move (the records of what the allocations hold are voided), the liquid
    moves (temporal reuse inside an unchanged window), the cuts move back -- every frame's window equals the single-rank pre-pass of that
    frame's liquid."""
    dev = torch.device("cuda:0")
    world, axis = 3, 0
    sA = scenes.to_device(scenes.sphere(64, 4, radius=0.30, device="cpu"), dev)
    sB = scenes.to_device(scenes.sphere(64, 4, radius=0.34, device="cpu"), dev)
    refs = {"A": _reference_arrays(sA), "B": _reference_arrays(sB)}
    cuts1 = np.asarray([0, 20, 44, 64], np.int32)
    cuts2 = np.asarray([0, 28, 36, 64], np.int32)
    frames = [("A", cuts1), ("B", cuts1), ("B", cuts2), ("A", cuts2), ("A", cuts1), ("B", cuts1)]
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    barrier = threading.Barrier(world)
    shared = {}
    lock = threading.Lock()

    def make_allreduce(rank):
        def allreduce(ptr, count, stream):
            assert hip.hipStreamSynchronize(C.c_void_p(stream)) == 0
            mine = np.empty(count, np.int32)
            assert hip.hipMemcpy(mine.ctypes.data, C.c_void_p(ptr), count * 4, 2) == 0
            barrier.wait()
            if rank == 0:
                shared["sum"] = np.zeros(count, np.int64)
            barrier.wait()
            with lock:
                shared["sum"] += mine
            barrier.wait()
            tot = shared["sum"].astype(np.int32)
            assert hip.hipMemcpy(C.c_void_p(ptr), tot.ctypes.data, count * 4, 1) == 0
            barrier.wait()
        return allreduce

    def rank_fn(r):
        pp = DevicePrepass(sA.res, sA.dx, sA.levels)
        ar = make_allreduce(r)
        for which, cuts in frames:
            sc = sA if which == "A" else sB
            ref, counts = refs[which]
            pp.set_slab(axis, cuts, r, ar)
            info = pp.run(sc.liquid, sc.solid)
            assert (info.levels, info.n_velocity, info.n_edge, info.n_center, info.n_regular) == counts, (r, which)
            lo, hi, _ = pp.window()
            _compare_window(pp, ref, info.levels, axis, lo, hi, sc.res)
        pp.close()
        return True

    _run_threads(world, rank_fn)


def _plan_arrays(s):
    sz = s.plan_sizes
    ti, tb = s.overlap_tiles
    a = dict(own_global=np.empty(sz.n_own, np.int32), row_ptr_local=np.empty(sz.n_own + 1, np.int32), col_local=np.empty(sz.nnz_local, np.int32),
             send_idx=np.empty(sz.n_send, np.int32), peers=np.empty(sz.n_peers, np.int32), send_counts=np.empty(sz.n_peers, np.int32),
             recv_counts=np.empty(sz.n_peers, np.int32), tiles_interior=np.empty(ti, np.int32), tiles_boundary=np.empty(tb, np.int32))
    capi.check(s.lib.avs_dist_get_plan_arrays(s.h, *[a[k].ctypes.data for k in ("own_global", "row_ptr_local", "col_local", "send_idx", "peers",
                                                                                  "send_counts", "recv_counts", "tiles_interior", "tiles_boundary")]))
    a["sizes"] = (sz.n_own, sz.n_halo, sz.nnz_local, sz.n_send)
    return a


@pytest.mark.parametrize("scene,world,axis,brick", [("beam64_L3", 2, 0, 0), ("beam64_L3", 4, 0, 0), ("sphere64_L4", 3, 2, 0), ("tank64_L3", 3, 0, 0),
                                                    ("noncubic", 4, 0, 0), ("varvisc128_L4", 4, 0, 0), ("varvisc128_L4", 8, 0, 1),
                                                    ("sheet128_L4", 4, 1, 1), ("beam64_L3", 2, 1, 1)])
def test_slab_local_assembly_equals_the_replicated_index_assembly(scene, world, axis, brick, built_lib, monkeypatch):
    """Same cuts -> same local systems (plan arrays equal entry for entry, the solve's solution bit for bit), with the native
    all-reduce of the in-process group (avs_dist_bind_prepass)."""
    dev = torch.device("cuda:0")
    if brick:
        monkeypatch.setenv("AVS_BRICK", "1")
    sc = scenes.to_device(SCENES[scene]("cpu"), dev)
    lib = capi.load()
    tol = 1e-10
    pp0 = DevicePrepass(sc.res, sc.dx, sc.levels)
    lv = pp0.run(sc.liquid, sc.solid).levels   # (the levels the grid and the data leave, oct.cpp:32-40, 198-211)
    pp0.close()

    def group_run(slab_cuts):
        grp = C.c_void_p()
        capi.check(lib.avs_local_group_create(world, C.byref(grp)))
        pps = [DevicePrepass(sc.res, sc.dx, sc.levels) for _ in range(world)]
        levels = {}
        if slab_cuts is None:   # the whole pre-pass on every rank
            for pp in pps:
                levels[0] = pp.run(sc.liquid, sc.solid).levels
        solvers = []

        def rank_fn(r):
            # (a host that does not know the capped level count creates its context after the run; here it is known)
            s = ViscositySolve(sc.res, sc.dx, sc.dt, lv, device=0)
            s.dist_init_local(grp, r)
            if slab_cuts is not None:
                s.dist_bind_prepass(pps[r], slab_cuts, axis)
                assert pps[r].run(sc.liquid, sc.solid).levels == lv
            else:
                assert levels[0] == lv
            pps[r].apply(s)
            s.set_scene_fields(sc)
            ai = s.dist_assemble(axis)
            plan = _plan_arrays(s)
            info = s.dist_solve(tol, 5000)
            x = s.dist_solution()
            ax, cuts = s.dist_cuts(world, 0)
            _, nxt = s.dist_cuts(world, 1)
            solvers.append(s)
            return dict(ai=ai, plan=plan, info=info, x=x, cuts=cuts, next_cuts=nxt, axis=ax, window=pps[r].window() if slab_cuts is not None else None)

        out = _run_threads(world, rank_fn)
        for s in solvers:
            s.close()
        for pp in pps:
            pp.close()
        lib.avs_local_group_destroy(grp)
        return out

    rep = group_run(None)
    cuts = rep[0]["cuts"]
    assert all(np.array_equal(r["cuts"], cuts) for r in rep) and rep[0]["axis"] == axis
    loc = group_run(cuts)
    n = len(rep[0]["x"])
    for r in range(world):
        a, b = rep[r], loc[r]
        assert a["plan"]["sizes"] == b["plan"]["sizes"], (r, a["plan"]["sizes"], b["plan"]["sizes"])
        for k in ("row_ptr_local", "col_local", "send_idx", "peers", "send_counts", "recv_counts", "tiles_interior", "tiles_boundary"):
            assert np.array_equal(a["plan"][k], b["plan"][k]), (r, k)
        assert a["ai"].nnz == b["ai"].nnz
        assert a["info"].iterations == b["info"].iterations and b["info"].converged == 1
        assert np.array_equal(a["x"], b["x"]), r            # same local systems, same transport: the same bits
        assert np.array_equal(b["next_cuts"], loc[0]["next_cuts"])
    # every DOF is owned exactly once, by reference id
    owned = np.concatenate([loc[r]["plan"]["own_global"] for r in range(world)])
    assert len(owned) == n and np.array_equal(np.sort(owned), np.arange(n, dtype=np.int32))
    # the work of a rank is its window's: with thin slabs no rank sweeps the whole octree
    if world >= 4:
        assert max(l["window"][2][0] for l in loc) < n


@pytest.mark.parametrize("scene,world,axis", [("beam64_L3", 2, 0), ("beam64_L3", 4, 1), ("sphere64_L4", 3, 2), ("tank64_L3", 3, 0), ("varvisc128_L4", 4, 0),
                                              ("sheet128_L4", 4, 1), ("noncubic", 2, 0)])
def test_slab_local_transfer_equals_the_whole_transfer_on_the_slab(scene, world, axis, built_lib):
    """The whole frame slab-local: pre-pass of the window, assembly, partitioned solve, gathered solution, and the post-solve transfer of the
    rank's window -- out of place and in place.  Reference: ONE context with the whole pyramid, handed the very same solution vector
    (avs_set_solution).  On the faces of its slab a rank's result equals the whole transfer bit for bit; every other face keeps the input."""
    dev = torch.device("cuda:0")
    sc = scenes.to_device(SCENES[scene]("cpu"), dev)
    lib = capi.load()
    pp0 = DevicePrepass(sc.res, sc.dx, sc.levels)
    lv = pp0.run(sc.liquid, sc.solid).levels
    cuts = _uniform_cuts(sc.res[axis], world)
    grp = C.c_void_p()
    capi.check(lib.avs_local_group_create(world, C.byref(grp)))
    keep = []

    def rank_fn(r):
        s = ViscositySolve(sc.res, sc.dx, sc.dt, lv, device=0)
        s.dist_init_local(grp, r)
        pp = DevicePrepass(sc.res, sc.dx, sc.levels)
        s.dist_bind_prepass(pp, cuts, axis)
        assert pp.run(sc.liquid, sc.solid).levels == lv
        pp.apply(s)
        s.set_scene_fields(sc)
        s.dist_assemble(axis)
        info = s.dist_solve(1e-8, 5000)
        assert info.converged == 1
        x = s.dist_solution()                       # gathers the whole vector on every rank (and keeps it in the context)
        outs = s.transfer_to_regular_grid()
        outs2 = s.transfer_to_regular_grid()        # a second transfer runs on the first one's temporal records
        vel = [v.clone() for v in sc.velocity]
        s.transfer_to_regular_grid_in_place(vel)
        keep.append((s, pp))
        return x, outs, outs2, [v.cpu().numpy() for v in vel]

    res = _run_threads(world, rank_fn)
    x = res[0][0]
    assert all(np.array_equal(x, r[0]) for r in res)
    ref = ViscositySolve(sc.res, sc.dx, sc.dt, lv, device=0)
    pp0.apply(ref)
    ref.set_scene_fields(sc)
    ref.assemble()
    ref.set_solution(x)
    want = ref.transfer_to_regular_grid()
    vin = [v.cpu().numpy() for v in sc.velocity]
    n_ax = sc.res[axis]
    changed = 0
    for r in range(world):
        lo, hi = int(cuts[r]), int(cuts[r + 1])
        for a in range(3):
            ext = want[a].shape[2 - axis]
            e = ext if r == world - 1 else hi          # (the last rank takes the face lattice's extra entry along the cut axis)
            inside = [slice(None)] * 3
            inside[2 - axis] = slice(lo, e)
            inside = tuple(inside)
            for got in (res[r][1][a], res[r][2][a], res[r][3][a]):
                assert np.array_equal(got[inside], want[a][inside]), (r, a)
                mask = np.ones(got.shape, bool)
                mask[inside] = False
                assert np.array_equal(got[mask], vin[a][mask]), (r, a, "outside the slab")
            changed += int(np.count_nonzero(want[a][inside] != vin[a][inside]))
    assert changed > 0 and n_ax > 0
    for s, pp in keep:
        s.close()
        pp.close()
    ref.close()
    pp0.close()
    lib.avs_local_group_destroy(grp)


def test_slab_local_assembly_at_512(built_lib):
    """The headline scene (512^3 fat beam, 7.4 M rows) cut four ways: the slab-local path -- four pre-pass objects, each on its window -- gives
    every rank the plan arrays of the replicated-index assembly, entry for entry, and 60 iterations of the partitioned loop give the same bits."""
    dev = torch.device("cuda:0")
    sc = scenes.fat_beam(512, 4, device=dev)
    world, axis = 4, 0
    lib = capi.load()

    def group_run(slab_cuts):
        grp = C.c_void_p()
        capi.check(lib.avs_local_group_create(world, C.byref(grp)))
        pps = [DevicePrepass(sc.res, sc.dx, sc.levels) for _ in range(world if slab_cuts is not None else 1)]
        if slab_cuts is None:
            assert pps[0].run(sc.liquid, sc.solid).levels == sc.levels
        solvers = []

        def rank_fn(r):
            s = ViscositySolve(sc.res, sc.dx, sc.dt, sc.levels, device=0)
            s.dist_init_local(grp, r)
            pp = pps[r] if slab_cuts is not None else pps[0]
            if slab_cuts is not None:
                s.dist_bind_prepass(pp, slab_cuts, axis)
                info = pp.run(sc.liquid, sc.solid)
                assert info.levels == sc.levels
            pp.apply(s)
            s.set_scene_fields(sc)
            ai = s.dist_assemble(axis)
            plan = _plan_arrays(s)
            info = s.dist_solve(1e-30, 60)     # (never converges: 60 iterations of a deterministic loop)
            x = s.dist_solution()
            _, cuts = s.dist_cuts(world, 0)
            vel = None
            if slab_cuts is not None:   # the transfer of the rank's slab, in place (rows of 64 faces from the slab's first face)
                vel = [v.clone() for v in sc.velocity]
                s.transfer_to_regular_grid_in_place(vel)
            solvers.append(s)
            return dict(nnz=ai.nnz, plan=plan, it=info.iterations, x=x, cuts=cuts, vel=vel, window=pp.window() if slab_cuts is not None else None)

        out = _run_threads(world, rank_fn)
        for s in solvers:
            s.close()
        for pp in pps:
            pp.close()
        lib.avs_local_group_destroy(grp)
        torch.cuda.empty_cache()
        return out

    rep = group_run(None)
    loc = group_run(rep[0]["cuts"])
    n = len(rep[0]["x"])
    for r in range(world):
        a, b = rep[r], loc[r]
        assert a["plan"]["sizes"] == b["plan"]["sizes"] and a["nnz"] == b["nnz"] and a["it"] == b["it"] == 60
        for k in ("row_ptr_local", "col_local", "send_idx", "peers", "send_counts", "recv_counts", "tiles_interior", "tiles_boundary"):
            assert np.array_equal(a["plan"][k], b["plan"][k]), (r, k)
        assert np.array_equal(a["x"], b["x"]), r
        assert b["window"][2][0] < 0.45 * n      # a rank's window holds well under half of the DOFs (a quarter + the halo)
    owned = np.concatenate([loc[r]["plan"]["own_global"] for r in range(world)])
    assert len(owned) == n and np.array_equal(np.sort(owned), np.arange(n, dtype=np.int32))
    # the transfer: ONE context with the whole pyramid, handed the same vector, against every rank's slab
    pp0 = DevicePrepass(sc.res, sc.dx, sc.levels)
    pp0.run(sc.liquid, sc.solid)
    ref = ViscositySolve(sc.res, sc.dx, sc.dt, sc.levels, device=0)
    pp0.apply(ref)
    ref.set_scene_fields(sc)
    ref.assemble()
    ref.set_solution(loc[0]["x"])
    want = [v.clone() for v in sc.velocity]
    ref.transfer_to_regular_grid_in_place(want)
    cuts = rep[0]["cuts"]
    changed = 0
    for r in range(world):
        lo, hi = int(cuts[r]), int(cuts[r + 1])
        for a in range(3):
            e = want[a].shape[2] if r == world - 1 else hi
            got = loc[r]["vel"][a]
            assert torch.equal(got[:, :, lo:e], want[a][:, :, lo:e]), (r, a)
            assert torch.equal(got[:, :, :lo], sc.velocity[a][:, :, :lo]) and torch.equal(got[:, :, e:], sc.velocity[a][:, :, e:]), (r, a, "outside the slab")
            changed += int((want[a][:, :, lo:e] != sc.velocity[a][:, :, lo:e]).sum())
    assert changed > 1000000
    ref.close()
    pp0.close()
