"""AVS_OPTION_FUSED_VECTOR_UPDATE (round 6): the two vector kernels of an iteration of the single-GPU launch-per-phase loop as ONE launch
with a grid barrier (k_update_fused, csrc/avs_pcg.hip).  It forms every sum in the order of the kernels it replaces, so the option is a pure
speed switch: same iteration count, same error, same solution BITS; a timed-out barrier redoes the solve with the two launches."""
import numpy as np
import pytest
import torch

from adaptiveviscositysolver_amd import DevicePrepass, ViscositySolve, capi, scenes

pytestmark = pytest.mark.gpu


def _solver(sc, probe=False):
    pp = DevicePrepass(sc.res, sc.dx, sc.levels, device=0)
    pinfo = pp.run(sc.liquid, sc.solid)
    s = ViscositySolve(sc.res, sc.dx, sc.dt, pinfo.levels, device=0, probe=probe)
    pp.apply(s)
    s.set_scene_fields(sc)
    pp.close()
    s.set_solver_option(capi.OPTION_RESIDENT_LOOP, 0)   # the launch-per-phase loop also where the system would fit the chip
    s.assemble()
    return s


CASES = {
    "beam256_L4_uniform": lambda dev: scenes.fat_beam(256, 4, device=dev),                            # 1.27 M rows, one small dictionary: coded diagonal
    "beam256_L4_mu_of_x": lambda dev: scenes.fat_beam(256, 4, variable_viscosity=True, device=dev),   # tile-local tables: 8-B inverse diagonal
    "sphere256_L4": lambda dev: scenes.sphere(256, 4, device=dev),
}


@pytest.mark.parametrize("name", list(CASES))
def test_fused_vector_update_changes_no_bit(name, built_lib):
    dev = torch.device("cuda:0")
    s = _solver(CASES[name](dev))
    n = int(s.ainfo.n_velocity)
    if n < 524_288:
        pytest.skip(f"{n} rows: below the fused launch's range")
    out = {}
    for mode in (0, 1, 0, 1):
        s.set_solver_option(capi.OPTION_FUSED_VECTOR_UPDATE, mode)
        for tol, cap in ((1e-3, 5000), (1e-9, 5000), (1e-12, 33), (1e-12, 64)):   # converged early / late, stopped by an odd / even cap
            info = s.solve(tol, cap)
            assert info.resident == 0
            f = s.matrix_format()
            assert f.fused_vector_update == mode and f.fused_vector_faults == 0
            key = (tol, cap)
            cur = (info.iterations, info.converged, info.error, s.solution())
            if key in out:
                ref = out[key]
                assert cur[0] == ref[0] and cur[1] == ref[1] and cur[2] == ref[2], (name, mode, key, cur[:3], ref[:3])
                assert np.array_equal(cur[3], ref[3]), (name, mode, key)
            else:
                out[key] = cur
    s.close()


def test_fused_vector_update_graph_and_plain_launches_agree(built_lib):
    dev = torch.device("cuda:0")
    s = _solver(scenes.fat_beam(256, 4, device=dev))
    s.set_solver_option(capi.OPTION_FUSED_VECTOR_UPDATE, 1)
    res = {}
    for g in (1, 0):
        s.set_solver_option(capi.OPTION_GRAPH_REPLAY, g)
        info = s.solve(1e-8, 5000)
        assert s.matrix_format().fused_vector_update == 1
        res[g] = (info.iterations, s.solution())
    assert res[0][0] == res[1][0] and res[0][0] > 64
    assert np.array_equal(res[0][1], res[1][1])
    s.close()


def test_timed_out_barrier_redoes_the_solve(built_lib, monkeypatch):
    """Probe build: AVS_PCG_FUSED_FAKE_FAULT makes the host treat the solve as if a fused launch's barrier had timed out -- the solve is
    redone from the initial guess with the two launches, the context stays on them, the answer is the same."""
    dev = torch.device("cuda:0")
    s = _solver(scenes.fat_beam(256, 4, device=dev), probe=True)
    s.set_solver_option(capi.OPTION_FUSED_VECTOR_UPDATE, 0)
    info0 = s.solve(1e-8, 5000)
    x0 = s.solution()
    s.set_solver_option(capi.OPTION_FUSED_VECTOR_UPDATE, 1)
    monkeypatch.setenv("AVS_PCG_FUSED_FAKE_FAULT", "1")
    info1 = s.solve(1e-8, 5000)
    f = s.matrix_format()
    assert f.fused_vector_faults == 1 and f.fused_vector_update == 0
    assert info1.iterations == info0.iterations and np.array_equal(s.solution(), x0)
    monkeypatch.delenv("AVS_PCG_FUSED_FAKE_FAULT")
    info2 = s.solve(1e-8, 5000)          # the context keeps the two launches
    assert s.matrix_format().fused_vector_update == 0 and info2.iterations == info0.iterations
    s.close()
