"""Multi-GPU code path on ONE GPU: virtual ranks (one avs_ctx + one host thread each, in-process
transport) run avs_dist_partition / avs_dist_solve and must reproduce the single-rank solve.
The only thing not exercised is the RCCL transport itself (world_size-1 init is checked)."""
import ctypes as C
import threading

import numpy as np
import pytest
import torch

from adaptiveviscositysolver_amd import ViscositySolve, capi, scenes
from util import build_pyramid, feed, rel_l2

pytestmark = pytest.mark.gpu


def make_solver(sc, pyr):
    s = ViscositySolve(sc.res, sc.dx, sc.dt, pyr.levels, device=0)
    feed(s, pyr)
    s.set_scene_fields(sc)
    for a in range(3):   # every regular face a DOF candidate: enough for the transfer smoke check below
        s.set_regular_index_field(a, np.where(pyr.vidx[0][a] == -3, -1, 0).astype(np.int32))
    s.assemble()
    return s


@pytest.mark.parametrize("world,cut_axis", [(2, -1), (4, 0), (3, 2)])
def test_virtual_ranks_match_single_solve(world, cut_axis, built_lib):
    dev = torch.device("cuda:0")
    sc = scenes.fat_beam(64, 3, variable_viscosity=True, device=dev)
    pyr = build_pyramid(sc)
    ref = make_solver(sc, pyr)
    tol = 1e-10
    iref = ref.solve(tol, 5000)
    xref = ref.solution()
    lib = capi.load()
    grp = C.c_void_p()
    capi.check(lib.avs_local_group_create(world, C.byref(grp)))
    solvers = [make_solver(sc, pyr) for _ in range(world)]
    results, errors, tiles, transfers = [None] * world, [], [None] * world, []

    def run(r):
        try:
            s = solvers[r]
            s.dist_init_local(grp, r)
            sz = s.dist_partition(cut_axis)
            info = s.dist_solve(tol, 5000)
            x = s.dist_solution()
            if r == 0:   # the gathered solution feeds the post-solve transfer on any rank
                transfers.append(s.transfer_to_regular_grid())
            results[r] = (info.iterations, info.converged, info.error, x, sz.n_own, sz.n_halo, sz.n_peers)
            tiles[r] = s.overlap_tiles
        except Exception as e:  # pragma: no cover
            errors.append((r, e))

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    assert not errors, errors
    assert all(r is not None for r in results)
    n_total = sum(r[4] for r in results)
    assert n_total == len(xref)
    for it, conv, err, x, n_own, n_halo, n_peers in results:
        assert conv == 1 and err <= tol
        assert abs(it - iref.iterations) <= 3
        assert rel_l2(x, xref) < 1e-8
        assert n_halo > 0 and n_peers >= 1
        assert n_halo < 0.5 * n_own          # slabs: the halo is a surface term
    assert len({r[0] for r in results}) == 1  # every rank reports the same iteration count
    # virtual ranks share one device (and its few hardware queues): they keep the host-mediated transport; the direct
    # transport is exercised with one PROCESS per rank below (test_processes_direct_transport)
    ci = solvers[0].dist_comm_info()
    assert ci["transport"] == "rccl" and ci["rccl_ranks"] == 0
    ref_out = ref.transfer_to_regular_grid()
    for a in range(3):
        assert np.allclose(transfers[0][a], ref_out[a], rtol=0, atol=1e-6 * max(1.0, float(np.abs(ref_out[a]).max())))
    for ti, tb in tiles:                      # the halo exchange overlaps the tiles that read no halo column
        assert tb >= 1 and ti + tb == -(-results[tiles.index((ti, tb))][4] // lib.avs_spmv_tile_rows())
    for s in solvers:
        s.close()
    lib.avs_local_group_destroy(grp)


def test_rccl_world_size_one(built_lib):
    """RCCL transport with a single rank: communicator creation, partition, solve."""
    dev = torch.device("cuda:0")
    sc = scenes.fat_beam(32, 3, device=dev)
    pyr = build_pyramid(sc)
    s = make_solver(sc, pyr)
    ref = s.solve(1e-10, 5000)
    xref = s.solution()
    lib = capi.load()
    buf = (C.c_uint8 * capi.UNIQUE_ID_BYTES)()
    capi.check(lib.avs_dist_get_unique_id(buf))
    capi.check(lib.avs_dist_init(s.h, buf, 0, 1))
    sz = s.dist_partition()
    assert sz.n_halo == 0 and sz.n_peers == 0 and sz.n_own == len(xref)
    info = s.dist_solve(1e-10, 5000)
    assert abs(info.iterations - ref.iterations) <= 1
    assert rel_l2(s.dist_solution(), xref) < 1e-8
    ci = s.dist_comm_info()
    assert ci["rccl_ranks"] == 1 and ci["transport"] == "direct" and ci["rccl_calls_per_iteration"] == 0
    assert ci["launches_per_iteration"] == 2 and ci["graph_replay"]      # update (+ push), SpMV (+ finalizer block)


@pytest.mark.parametrize("scene,world", [("beam", 2), ("varvisc", 2), ("beam128", 3), ("beam128", 8), ("varvisc128", 3),
                                         ("beam128_brick", 3), ("beam128L4_brick", 2), ("varvisc128_brick", 2)])   # round 5: the brick-structured form (+ its value-code variant)
def test_processes_direct_transport(scene, world, tmp_path, built_lib):
    """One PROCESS per rank (both on cuda:0): comm blocks mapped through HIP IPC handles, halo entries stored straight into
    the neighbour's block, CG sums by flag-based all-gather -- no RCCL anywhere (hosted group, blobs through files)."""
    import os
    import subprocess
    import sys
    dev = torch.device("cuda:0")
    sc = {"beam": lambda: scenes.fat_beam(64, 3, device=dev),
          "varvisc": lambda: scenes.fat_beam(64, 3, variable_viscosity=True, device=dev),
          "beam128": lambda: scenes.fat_beam(128, 3, device=dev),
          "beam128_brick": lambda: scenes.fat_beam(128, 3, device=dev),
          "beam128L4_brick": lambda: scenes.fat_beam(128, 4, device=dev),
          "varvisc128_brick": lambda: scenes.fat_beam(128, 4, variable_viscosity=True, device=dev),
          # tile-local dictionaries + windowed columns through the HALO instantiation of the SpMV (peer-written halo area)
          "varvisc128": lambda: scenes.fat_beam(128, 4, variable_viscosity=True, device=dev)}[scene]()
    pyr = build_pyramid(sc)
    ref = make_solver(sc, pyr)
    tol = 1e-9
    iref = ref.solve(tol, 5000)
    xref = ref.solution()
    ref.close()
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", AVS_DIST_TIMEOUT_MS="30000")
    if scene.endswith("_brick"):
        env["AVS_BRICK"] = "1"   # (slabs this small would not get the form by the size rule)
    procs = [subprocess.Popen([sys.executable, os.path.join(here, "hosted_rank.py"), str(tmp_path), str(r), str(world), scene, repr(tol)],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = [p.communicate(timeout=280) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    x = np.zeros_like(xref)
    n_own = 0
    for r in range(world):
        x += np.load(tmp_path / f"x_{r}.npy")
        brick_tiles, brick_rows = np.load(tmp_path / f"fmt_{r}.npy")
        assert (brick_tiles > 0) == scene.endswith("_brick"), (scene, r, brick_tiles)
        it1, c1, it2, c2, own, halo, direct, rccl_calls, launches, tile_tables, windows, resident, st_rounds, st_bad, paranoid = np.load(tmp_path / f"info_{r}.npy")
        assert resident == 0     # ranks that share one GPU keep the launch-per-phase loop (the resident loop needs every CU)
        assert st_rounds == 64 and st_bad == 0 and paranoid == 0    # the transport self-test ran over the mapped blocks and passed
        if scene == "varvisc128":
            assert tile_tables == 1 and windows == 1
        assert c1 == 1 and c2 == 1 and it1 == it2 and abs(it1 - iref.iterations) <= 3
        assert direct == 1 and rccl_calls == 0 and launches <= 4 and halo > 0
        n_own += int(own)
    assert n_own == len(xref)
    assert rel_l2(x, xref) < 1e-7


def _run_hosted(tmp_path, scene, world, tol, extra_env):
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", AVS_DIST_TIMEOUT_MS="30000")
    env.update(extra_env)
    procs = [subprocess.Popen([sys.executable, os.path.join(here, "hosted_rank.py"), str(tmp_path), str(r), str(world), scene, repr(tol)],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = [p.communicate(timeout=280) for p in procs]
    return [(p.returncode, so, se) for p, (so, se) in zip(procs, outs)]


def test_processes_paranoid_mode(tmp_path, built_lib):
    """AVS_DIST_PARANOID=1: every round's halo segments are re-added by the reader and compared with the checksum the sender left
    ahead of its flag (round-2 review: "make the guard able to see a stale halo").  Same iterations and solution as without."""
    dev = torch.device("cuda:0")
    sc = scenes.fat_beam(128, 3, device=dev)
    ref = make_solver(sc, build_pyramid(sc))
    tol = 1e-9
    iref = ref.solve(tol, 5000)
    xref = ref.solution()
    ref.close()
    res = _run_hosted(tmp_path, "beam128", 3, tol, {"AVS_DIST_PARANOID": "1"})
    for rc, so, se in res:
        assert rc == 0, se[-3000:]
    x = np.zeros_like(xref)
    for r in range(3):
        x += np.load(tmp_path / f"x_{r}.npy")
        info = np.load(tmp_path / f"info_{r}.npy")
        assert info[1] == 1 and abs(info[0] - iref.iterations) <= 3
        assert info[-3] == 64 and info[-2] == 0 and info[-1] == 1 and info[-4] == 0   # (paranoid mode keeps the launch-per-phase loop)
    assert rel_l2(x, xref) < 1e-7


@pytest.mark.parametrize("scene,world,cus", [("beam128", 2, 96), ("beam128", 3, 64),
                                             ("beam128", 2, 40)])   # 40 CUs per rank: the slab exceeds their register files -> streamed rows
def test_processes_resident_loop_across_ranks(scene, world, cus, tmp_path, built_lib):
    """The CU-resident loop (avs_pcg_resident.inl) between REAL ranks: one process per rank on a share of one GPU's CUs
    (AVS_CG_RESIDENT_CUS; on a multi-GPU node every rank has all CUs of its own GPU).  Pushes into the peer's halo area, halo flags
    raised by the grid barrier's last arriver, rank sums all-gathered by workgroup 0 through the comm blocks -- against the
    single-GPU solve."""
    dev = torch.device("cuda:0")
    sc = scenes.fat_beam(128, 3, device=dev)
    ref = make_solver(sc, build_pyramid(sc))
    tol = 1e-9
    iref = ref.solve(tol, 5000)
    xref = ref.solution()
    ref.close()
    res = _run_hosted(tmp_path, scene, world, tol, {"AVS_CG_RESIDENT_CUS": str(cus), "AVS_DIST_TIMEOUT_MS": "8000"})
    for rc, so, se in res:
        assert rc == 0, se[-3000:]
    x = np.zeros_like(xref)
    for r in range(world):
        x += np.load(tmp_path / f"x_{r}.npy")
        info = np.load(tmp_path / f"info_{r}.npy")
        assert info[1] == 1 and info[3] == 1 and abs(info[0] - iref.iterations) <= 3 and info[0] == info[2]
        assert info[-4] == 1, "the resident loop did not run"
    assert rel_l2(x, xref) < 1e-7


@pytest.mark.parametrize("stale_round,where", [(10, "self-test"), (200, "solve")])
def test_stale_halo_entry_is_detected(stale_round, where, tmp_path, built_lib):
    """Test hook AVS_DIST_INJECT_STALE=k: in round k every rank does NOT store the first entry it owes its first peer, so that peer
    multiplies with the value of round k-1 -- exactly what a flag overtaking its data over xGMI would produce.  Rounds 1..64 are the
    transport self-test (must fail the connect), later ones belong to the solve (must stop with AVS_ERCCL in paranoid mode)."""
    res = _run_hosted(tmp_path, "beam128", 2, 1e-9, {"AVS_DIST_INJECT_STALE": str(stale_round), "AVS_DIST_TIMEOUT_MS": "4000"})
    assert all(rc == 7 for rc, _, _ in res), [se[-500:] for _, _, se in res]
    msgs = [open(tmp_path / f"err_{r}.txt").read() for r in range(2)]
    assert all(m.split()[0] == str(capi.ERCCL) for m in msgs), msgs
    if where == "self-test":
        assert all("self-test failed" in m for m in msgs), msgs
    else:
        assert any("checksum" in m for m in msgs), msgs


def _plan_arrays(s):
    lib = s.lib
    sz = s.plan_sizes
    ti, tb = s.overlap_tiles
    arr = lambda n: np.empty(int(n), np.int32)
    own, rpl, cl, sidx = arr(sz.n_own), arr(sz.n_own + 1), arr(sz.nnz_local), arr(sz.n_send)
    peers, sc_, rc = arr(sz.n_peers), arr(sz.n_peers), arr(sz.n_peers)
    tint, tbnd = arr(ti), arr(tb)
    capi.check(lib.avs_dist_get_plan_arrays(s.h, *[a.ctypes.data for a in (own, rpl, cl, sidx, peers, sc_, rc, tint, tbnd)]))
    return dict(own=own, row_ptr=rpl, col=cl, send_idx=sidx, peers=peers, send_counts=sc_, recv_counts=rc,
                tiles_int=tint, tiles_bnd=tbnd, sizes=(sz.n_own, sz.n_halo, sz.nnz_local, sz.n_send, sz.n_peers))


@pytest.mark.parametrize("world,cut_axis,scene", [(2, -1, "beam"), (4, 0, "beam"), (3, 2, "sphere"), (8, 0, "beam")])
def test_device_planner_equals_host_planner(world, cut_axis, scene, built_lib, monkeypatch):
    """avs_dist_partition builds the plan on the device; AVS_DIST_PLAN=host runs avs_partition.cpp on a downloaded
    copy of the pattern.  Every array of every rank must be identical (integer work: bit-exact)."""
    dev = torch.device("cuda:0")
    sc = scenes.fat_beam(64, 3, device=dev) if scene == "beam" else scenes.sphere(64, 4, device=dev)
    pyr = build_pyramid(sc)
    s = make_solver(sc, pyr)
    lib = capi.load()
    for r in range(world):
        plans = []
        for mode in ("device", "host"):
            monkeypatch.setenv("AVS_DIST_PLAN", mode)
            s.set_solver_option(capi.OPTION_RELOAD_ENVIRONMENT, 1)   # (the environment is read at avs_create)
            grp = C.c_void_p()
            capi.check(lib.avs_local_group_create(world, C.byref(grp)))
            s.dist_init_local(grp, r)          # planning needs no peer: each rank plans from the replicated system
            s.dist_partition(cut_axis)
            plans.append(_plan_arrays(s))
            lib.avs_local_group_destroy(grp)
        dev_plan, host_plan = plans
        assert dev_plan["sizes"] == host_plan["sizes"], (r, dev_plan["sizes"], host_plan["sizes"])
        for k in ("own", "row_ptr", "col", "send_idx", "peers", "send_counts", "recv_counts", "tiles_int", "tiles_bnd"):
            assert np.array_equal(dev_plan[k], host_plan[k]), (r, k)
    s.close()


@pytest.mark.parametrize("world,cut_axis,scene", [(1, -1, "beam"), (2, -1, "beam"), (4, 0, "varvisc"), (3, 2, "sphere"), (3, 0, "varvisc128"),
                                                  (1, -1, "beam128_brick"), (3, 0, "beam128_brick"), (2, 2, "beam128L4_brick"), (2, 0, "varvisc128_brick")])
def test_distributed_assembly_matches_single_solve(world, cut_axis, scene, built_lib, monkeypatch):
    """avs_dist_assemble: every rank assembles only its own rows (no global matrix) -- the partitioned solve must
    reproduce the single-rank solve, the local systems must add up to the global one, and the send / receive lists of
    neighbouring ranks must agree (they are derived independently on each side from the symmetric pattern)."""
    dev = torch.device("cuda:0")
    if scene.endswith("_brick"):   # round 5: the brick-structured form of the rank's local rows ([owned | halo] columns, halo in the vector's tail)
        monkeypatch.setenv("AVS_BRICK", "1")
    sc = {"beam": lambda: scenes.fat_beam(64, 3, device=dev),
          "varvisc": lambda: scenes.fat_beam(64, 3, variable_viscosity=True, device=dev),
          "sphere": lambda: scenes.sphere(64, 4, device=dev),
          "beam128_brick": lambda: scenes.fat_beam(128, 3, device=dev),
          "beam128L4_brick": lambda: scenes.fat_beam(128, 4, device=dev),
          "varvisc128_brick": lambda: scenes.fat_beam(128, 4, variable_viscosity=True, device=dev),
          # tens of thousands of distinct values per rank: tile-local dictionaries + windowed columns on matrices with halo
          # columns and the [interior | halo-reading] row order (the form whose head-of-pass decode once gathered out of range)
          "varvisc128": lambda: scenes.fat_beam(128, 4, variable_viscosity=True, device=dev)}[scene]()
    pyr = build_pyramid(sc)
    ref = make_solver(sc, pyr)
    tol = 1e-10
    iref = ref.solve(tol, 5000)
    xref = ref.solution()
    nnz_ref = ref.info().nnz
    lib = capi.load()
    grp = C.c_void_p()
    capi.check(lib.avs_local_group_create(world, C.byref(grp)))
    solvers = []
    for _ in range(world):
        s = ViscositySolve(sc.res, sc.dx, sc.dt, pyr.levels, device=0)
        feed(s, pyr)
        s.set_scene_fields(sc)
        solvers.append(s)
    results, errors = [None] * world, []

    def run(r):
        try:
            s = solvers[r]
            s.dist_init_local(grp, r)
            ai = s.dist_assemble(cut_axis)
            plan = _plan_arrays(s)
            info = s.dist_solve(tol, 5000)
            x = s.dist_solution()
            if scene == "varvisc128":
                fmt = s.matrix_format()
                assert fmt.tile_local_tables == 1 and fmt.column_windows == 1 and fmt.bytes_per_nonzero == 4
            if scene.endswith("_brick"):
                fmt = s.matrix_format()
                assert fmt.brick_tiles > 0 and fmt.brick_pattern_rows > 0.6 * plan["sizes"][0], (r, fmt.brick_tiles, fmt.brick_pattern_rows)
                assert fmt.brick_value_codes == (1 if scene.startswith("varvisc") else 0)
            results[r] = (info, x, ai, plan)
        except Exception as e:  # pragma: no cover
            errors.append((r, e))

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=300)
    assert not errors, errors
    assert all(r is not None for r in results)
    assert sum(r[3]["sizes"][0] for r in results) == len(xref)          # every DOF owned exactly once
    assert sum(r[2].nnz for r in results) == nnz_ref                    # local rows add up to the global matrix
    for info, x, ai, plan in results:
        assert info.converged == 1 and info.error <= tol
        assert abs(info.iterations - iref.iterations) <= 3
        assert rel_l2(x, xref) < 1e-8
    # what r sends to q is what q expects from r
    for r in range(world):
        pr = results[r][3]
        for i, q in enumerate(pr["peers"]):
            pq = results[int(q)][3]
            j = list(pq["peers"]).index(r)
            assert pr["send_counts"][i] == pq["recv_counts"][j]
            assert pr["recv_counts"][i] == pq["send_counts"][j]
    with pytest.raises(capi.AvsError):     # no global matrix in this mode
        solvers[0].csr()
    for s in solvers:
        s.close()
    ref.close()
    lib.avs_local_group_destroy(grp)


def test_distributed_assembly_call_order(built_lib):
    """Call-sequence errors are reported, not crashed on (avs_status AVS_ESTATE)."""
    dev = torch.device("cuda:0")
    sc = scenes.fat_beam(32, 3, device=dev)
    pyr = build_pyramid(sc)
    s = ViscositySolve(sc.res, sc.dx, sc.dt, pyr.levels, device=0)
    feed(s, pyr)
    s.set_scene_fields(sc)
    with pytest.raises(capi.AvsError) as e:          # no communicator yet
        s.dist_assemble()
    assert e.value.status == capi.ESTATE
    lib = capi.load()
    grp = C.c_void_p()
    capi.check(lib.avs_local_group_create(1, C.byref(grp)))
    s.dist_init_local(grp, 0)
    with pytest.raises(capi.AvsError) as e:          # nothing assembled: no partitioned system to solve
        s.dist_solve(1e-6, 10)
    assert e.value.status == capi.ESTATE
    s.dist_assemble()
    for call in (s.solve, s.csr):   # the global matrix does not exist in this mode
        with pytest.raises(capi.AvsError) as e:
            call()
        assert e.value.status == capi.ESTATE
    info = s.dist_solve(1e-8, 2000)
    assert info.converged == 1
    s.assemble()                                     # back to the single-GPU path on the same context
    ref = s.solve(1e-8, 2000)
    assert abs(ref.iterations - info.iterations) <= 2
    s.close()
    lib.avs_local_group_destroy(grp)


def test_loopback_measurement_mode(built_lib, monkeypatch):
    """AVS_DIST_LOOPBACK=1 (tools/loopback_scaling.py): one rank of a 4-way partition alone on the GPU, peers looped back onto
    itself.  Not a correct solve (zero halo) -- only checked for what the tool relies on: the loop runs the requested number of
    iterations through the direct transport with 2 launches per iteration and leaves finite numbers."""
    monkeypatch.setenv("AVS_DIST_LOOPBACK", "1")
    dev = torch.device("cuda:0")
    sc = scenes.fat_beam(64, 3, device=dev)
    pyr = build_pyramid(sc)
    for rank in (0, 2):
        s = ViscositySolve(sc.res, sc.dx, sc.dt, pyr.levels, device=0)
        feed(s, pyr)
        s.set_scene_fields(sc)
        capi.check(s.lib.avs_dist_init_hosted(s.h, rank, 4))
        s.dist_assemble()
        capi.check(s.lib.avs_dist_import_blobs(s.h, None))
        info = s.dist_solve(1e-30, 96)
        ci = s.dist_comm_info()
        assert info.iterations == 96 and info.converged == 0 and np.isfinite(info.error)
        assert ci["transport"] == "direct" and ci["launches_per_iteration"] == 2 and ci["rccl_calls_per_iteration"] == 0
        ti, tb = s.overlap_tiles
        assert 0 < tb < ti            # [interior | halo-reading] row order: few halo-reading tiles
        s.close()


@pytest.mark.parametrize("world,rank,scene", [(2, 0, "beam128"), (3, 1, "beam128"), (2, 1, "sheet128")])
def test_local_brick_form_product(world, rank, scene, built_lib, monkeypatch):
    """the brick-structured form of ONE rank's local rows against the word stream of the same rows: a random [owned | halo] vector through
    both (plain and fused-dot instantiations) must give the same y bit for bit (the word stream is the form the other tests pin against
    the oracle).  Both plans use the plain ascending row order (AVS_DIST_SPLIT_ROWS=0), so the local numberings are the same."""
    dev = torch.device("cuda:0")
    sc = scenes.fat_beam(128, 4, device=dev) if scene == "beam128" else scenes.thin_sheet(128, 4, thickness_cells=12, device=dev)
    pyr = build_pyramid(sc)
    monkeypatch.setenv("AVS_DIST_SPLIT_ROWS", "0")
    rng = np.random.default_rng(5)
    ys, dots, owns, xs = {}, {}, {}, None
    for brick in (1, 0):
        monkeypatch.setenv("AVS_BRICK", str(brick))
        s = ViscositySolve(sc.res, sc.dx, sc.dt, pyr.levels, device=0, probe=True)
        feed(s, pyr)
        s.set_scene_fields(sc)
        capi.check(s.lib.avs_dist_init_hosted(s.h, rank, world))   # planning and assembly need no peer
        s.dist_assemble(0)
        sz = s.plan_sizes
        fmt = s.matrix_format()
        assert (fmt.brick_tiles > 0) == bool(brick) and sz.n_halo > 0
        n_ext = int(sz.n_own + sz.n_halo)
        if xs is None:
            xs = [torch.from_numpy(rng.standard_normal(n_ext) * 10.0 ** rng.integers(-2, 3, n_ext)).to(dev) for _ in range(2)]
        owns[brick] = _plan_arrays(s)["own"].copy()
        for t, x in enumerate(xs):
            assert len(x) == n_ext
            for fused in (0, 1):
                y = torch.full((int(sz.n_own),), float("nan"), dtype=torch.float64, device=dev)
                dot = C.c_double()
                capi.check(s.lib.avs_dist_spmv_local_form(s.h, x.data_ptr(), y.data_ptr(), fused, C.byref(dot)))
                ys[(brick, t, fused)] = y.cpu().numpy()
                dots[(brick, t, fused)] = dot.value
        s.close()
    assert np.array_equal(owns[0], owns[1])
    for t in range(2):
        for fused in (0, 1):
            yb, yw = ys[(1, t, fused)], ys[(0, t, fused)]
            assert np.array_equal(yb.view(np.int64), yw.view(np.int64)), (t, fused, int((yb != yw).sum()))
        assert abs(dots[(1, t, 1)] - dots[(0, t, 1)]) <= 1e-9 * max(1.0, abs(dots[(0, t, 1)]))
