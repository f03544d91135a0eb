"""CU-resident PCG (avs_pcg_resident.inl, the default for systems that fit the chip): one cooperative launch for all iterations, the
packed matrix words in the register files, the vector slices in LDS.  It must agree with the oracle like the launch-per-phase loops do (same recurrences as the direct
transport's single-reduction loop; only the order of additions inside the three dot products differs)."""
import os

import numpy as np
import pytest
import torch

from adaptiveviscositysolver_amd import ViscositySolve, capi, scenes
from util import build_pyramid, feed, oracle_from_pyramid, rel_l2

pytestmark = pytest.mark.gpu

CASES = {
    "beam64_L3_wall": lambda: scenes.fat_beam(64, 3, wall=True),
    "sphere64_L4": lambda: scenes.sphere(64, 4),
    "beam128_L3": lambda: scenes.fat_beam(128, 3),                       # BASELINE configs[1]
    "hip_buckling": lambda: scenes.viscous_buckling_scene(),            # 0.79 M rows: the vector slices + remote columns nearly fill the LDS
    "sphere_obstacle": lambda: scenes.sphere_with_obstacle(64, 4),        # curved solid boundary: rhs carries the boundary terms
    "viscous_beam_hip_coarse": lambda: scenes.viscous_beam_scene(coarsen=2),
}


@pytest.fixture
def resident_env():
    old = os.environ.get("AVS_CG_RESIDENT")
    os.environ["AVS_CG_RESIDENT"] = "1"
    yield
    if old is None:
        os.environ.pop("AVS_CG_RESIDENT", None)
    else:
        os.environ["AVS_CG_RESIDENT"] = old


@pytest.mark.parametrize("name", list(CASES))
def test_resident_single_gpu_solve_matches_oracle(name, resident_env, built_lib):
    sc = CASES[name]()
    dsc = scenes.to_device(sc, torch.device("cuda:0"))
    pyr = build_pyramid(dsc)
    s = ViscositySolve(sc.res, sc.dx, sc.dt, pyr.levels, device=0, field_res=sc.field_res)
    feed(s, pyr)
    s.set_scene_fields(scenes.crop_to_field(dsc))
    s.assemble()
    o = oracle_from_pyramid(sc, pyr)
    o.hot_path()
    for tol in (1e-10, 1e-3):
        info = s.solve(tol, 5000)
        assert info.resident == 1, "the resident loop did not run (system not eligible?)"
        xo, io = o.solve(tol, 5000)
        assert info.converged == 1 and info.error <= tol
        assert abs(info.iterations - io.iterations) <= max(3, io.iterations // 100), (info.iterations, io.iterations)
        if tol < 1e-6:
            assert rel_l2(s.solution(), xo) < 1e-7
    # the same context through the launch-per-phase loop: the two GPU loops agree even closer
    x_res = s.solution()
    s.set_solver_option(capi.OPTION_RESIDENT_LOOP, 0)
    info2 = s.solve(1e-3, 5000)
    assert info2.resident == 0
    assert rel_l2(s.solution(), x_res) < 1e-2 * 1e-3 * 50   # both within tol of the same solution
    s.close()


@pytest.mark.parametrize("which", ["tile_dictionaries", "density_tensor", "too_many_rows"])
def test_resident_refuses_what_does_not_fit(which, resident_env, built_lib):
    """Systems that are not in the packed single-dictionary form (a viscosity / density field with thousands of distinct values) or
    too large for the register files silently keep the launch-per-phase loop."""
    dev = torch.device("cuda:0")
    if which == "too_many_rows":    # 1.27 M rows: more than the register files hold; with streamed rows switched off the loop must decline
        os.environ["AVS_CG_RESIDENT_NO_STREAM"] = "1"
    try:
        _refuses(which, dev)
    finally:
        os.environ.pop("AVS_CG_RESIDENT_NO_STREAM", None)


def _refuses(which, dev):
    sc = {"tile_dictionaries": lambda: scenes.fat_beam(256, 4, variable_viscosity=True, device=dev),
          "density_tensor": lambda: scenes.with_sampled_fields(scenes.sphere_with_obstacle(64, 4, device=dev)),
          "too_many_rows": lambda: scenes.fat_beam(256, 5, device=dev)}[which]()
    pyr = build_pyramid(sc)
    s = ViscositySolve(sc.res, sc.dx, sc.dt, pyr.levels, device=0)
    feed(s, pyr)
    s.set_scene_fields(sc)
    s.assemble()
    info = s.solve(1e-3, 2500)
    assert info.resident == 0 and info.converged == 1
    s.close()


def test_resident_direct_transport_world_1(built_lib):
    """The partitioned path (direct transport, one rank): resident by default, against the single-GPU solve of the same system."""
    import ctypes as C
    sc = scenes.fat_beam(128, 4, device=torch.device("cuda:0"))
    pyr = build_pyramid(sc)
    s = ViscositySolve(sc.res, sc.dx, sc.dt, pyr.levels, device=0)
    feed(s, pyr)
    s.set_scene_fields(sc)
    buf = (C.c_uint8 * capi.UNIQUE_ID_BYTES)()
    capi.check(s.lib.avs_dist_get_unique_id(buf))
    capi.check(s.lib.avs_dist_init(s.h, buf, 0, 1))
    s.assemble()
    ref = s.solve(1e-9, 5000)
    xref = s.solution()
    s.dist_assemble()
    info = s.dist_solve(1e-9, 5000)
    assert info.resident == 1 and info.converged == 1 and abs(info.iterations - ref.iterations) <= 3
    assert rel_l2(s.dist_solution(), xref) < 1e-7
    info = s.dist_solve(1e-9, 5000)          # again: the plan is re-used
    assert info.resident == 1 and abs(info.iterations - ref.iterations) <= 3
    s.close()


def test_resident_long_row_path(resident_env, built_lib):
    """Rows that do not fit a lane's registers keep their first words there and read the rest from memory.  No scene here has a row
    of more than 75 merged entries, so the test lowers the quads a lane may use to 4 (20 words): every transition row takes the path."""
    os.environ["AVS_CG_RESIDENT_MAX_QUADS"] = "4"
    try:
        sc = scenes.sphere(64, 4)
        dsc = scenes.to_device(sc, torch.device("cuda:0"))
        pyr = build_pyramid(dsc)
        s = ViscositySolve(sc.res, sc.dx, sc.dt, pyr.levels, device=0)
        feed(s, pyr)
        s.set_scene_fields(dsc)
        s.assemble()
        rp = s.csr()[0]
        assert int(np.diff(rp).max()) > 20
        info = s.solve(1e-10, 5000)
        assert info.resident == 1 and info.converged == 1
        o = oracle_from_pyramid(sc, pyr)
        o.hot_path()
        xo, io = o.solve(1e-10, 5000)
        assert abs(info.iterations - io.iterations) <= 3 and rel_l2(s.solution(), xo) < 1e-7
        s.close()
    finally:
        os.environ.pop("AVS_CG_RESIDENT_MAX_QUADS", None)


@pytest.mark.parametrize("mode", ["forced_on_64_cus", "beam256_whole_chip"])
def test_resident_streamed_rows(mode, resident_env, built_lib):
    """A system larger than the register files: every lane keeps what fits and STREAMS the quads of its remaining rows from memory
    (lane-interleaved per wave), the row-local vectors r, p, s move to global memory as the LDS requires.  forced_on_64_cus: the
    128^3 beam on a quarter of the chip (34 % of the words streamed) with a 64 K-column bitmap chunk in the plan kernel (six passes);
    beam256_whole_chip: 1.27 M rows on all CUs (27 % streamed).  Against the oracle like every other loop."""
    env = {"AVS_CG_RESIDENT_CUS": "64", "AVS_CG_RESIDENT_REMAP_CHUNK": "65536"} if mode == "forced_on_64_cus" else {}
    os.environ.update(env)
    try:
        sc = scenes.fat_beam(128, 3) if mode == "forced_on_64_cus" else scenes.fat_beam(256, 4)
        dsc = scenes.to_device(sc, torch.device("cuda:0"))
        pyr = build_pyramid(dsc)
        s = ViscositySolve(sc.res, sc.dx, sc.dt, pyr.levels, device=0)
        feed(s, pyr)
        s.set_scene_fields(dsc)
        s.assemble()
        o = oracle_from_pyramid(sc, pyr)
        o.hot_path()
        for tol in (1e-10, 1e-3):
            info = s.solve(tol, 5000)
            assert info.resident == 1 and info.converged == 1 and info.error <= tol
            xo, io = o.solve(tol, 5000)
            assert abs(info.iterations - io.iterations) <= max(3, io.iterations // 100), (info.iterations, io.iterations)
            if tol < 1e-6:
                assert rel_l2(s.solution(), xo) < 1e-7
        x_res = s.solution()
        s.set_solver_option(capi.OPTION_RESIDENT_LOOP, 0)   # the same context through the launch-per-phase loop
        info2 = s.solve(1e-3, 5000)
        assert info2.resident == 0 and rel_l2(s.solution(), x_res) < 5e-4
        s.close()
    finally:
        for k in env:
            os.environ.pop(k, None)


def test_resident_plan_follows_a_reassembly(resident_env, built_lib):
    """The resident plan holds a re-encoding of the matrix WORDS.  A second assembly on the same context with the same DOF count
    (next frame, other viscosity / density) rewrites the same device buffers: the plan must be rebuilt (it is keyed on the value
    index's generation, not only on pointers).  The second system here has other values AND another dictionary order."""
    sc = scenes.fat_beam(128, 3)
    dsc = scenes.to_device(sc, torch.device("cuda:0"))
    pyr = build_pyramid(dsc)
    s = ViscositySolve(sc.res, sc.dx, sc.dt, pyr.levels, device=0)
    feed(s, pyr)
    s.set_scene_fields(dsc)
    s.assemble()
    assert s.solve(1e-9, 5000).resident == 1
    x1 = s.solution()
    s.set_field(capi.FIELD_VISCOSITY, 0, None, 3.0)          # viscous term ~3000x smaller against the same mass term
    s.set_field(capi.FIELD_DENSITY, 0, None, 1730.0)
    s.assemble()
    info = s.solve(1e-9, 5000)
    assert info.resident == 1 and info.converged == 1
    x2 = s.solution()
    s.set_solver_option(capi.OPTION_RESIDENT_LOOP, 0)
    ref = s.solve(1e-9, 5000)
    assert ref.resident == 0 and abs(ref.iterations - info.iterations) <= 2
    assert rel_l2(x2, s.solution()) < 1e-7
    assert rel_l2(x2, x1) > 1e-6                              # (it is another system)
    s.close()


def test_resident_fault_is_redone_by_the_launch_per_phase_loop(resident_env, monkeypatch, built_lib):
    """A bounded wait inside the cooperative launch that times out (a GPU shared with a viewport: the grid not co-resident in time) must
    not cost the frame: the same avs_solve call restores the initial guess and solves with the launch-per-phase loop, and the context stays
    on that loop.  AVS_CG_RESIDENT_FAKE_FAULT makes the host treat the launch as faulted (advisor, round 3)."""
    sc = scenes.fat_beam(64, 3)
    dsc = scenes.to_device(sc, torch.device("cuda:0"))
    pyr = build_pyramid(dsc)
    s = ViscositySolve(sc.res, sc.dx, sc.dt, pyr.levels, device=0, probe=True)   # (the hook exists in the probe build only)
    feed(s, pyr)
    s.set_scene_fields(dsc)
    s.assemble()
    good = s.solve(1e-8, 5000)
    assert good.resident == 1 and good.converged == 1
    x_good = np.array(s.solution(), copy=True)
    monkeypatch.setenv("AVS_CG_RESIDENT_FAKE_FAULT", "1")
    s.assemble()                       # a new plan, the resident loop is tried again ...
    info = s.solve(1e-8, 5000)         # ... "faults", and the call still returns the solution
    assert info.converged == 1 and info.resident == 0
    assert abs(info.iterations - good.iterations) <= max(3, good.iterations // 100)
    assert rel_l2(s.solution(), x_good) < 1e-7
    monkeypatch.delenv("AVS_CG_RESIDENT_FAKE_FAULT")
    again = s.solve(1e-8, 5000)        # the plan was retired: the context stays on the launch-per-phase loop
    assert again.converged == 1 and again.resident == 0
    s.close()
