"""avs_cancel: the user interrupt of the reference (UT_Interrupt::opInterrupt(), cpp:2528; HDK_OctreeGrid.cpp:584-588) -- a second
thread ends a running solve; the call returns AVS_OK with converged = 0, cancelled = 1, and the request is consumed."""
import threading
import time

import numpy as np
import pytest
import torch

from adaptiveviscositysolver_amd import ViscositySolve, capi, scenes
from util import build_pyramid, feed

pytestmark = pytest.mark.gpu


def _solver(monkeypatch, resident, big=False):
    monkeypatch.setenv("AVS_CG_RESIDENT", "1" if resident else "0")
    dev = torch.device("cuda:0")
    sc = scenes.fat_beam(256 if big else 128, 4 if big else 3, device=dev)
    pyr = build_pyramid(sc)
    s = ViscositySolve(sc.res, sc.dx, sc.dt, pyr.levels, device=0)
    feed(s, pyr)
    s.set_scene_fields(sc)
    s.assemble()
    return s


def test_cancel_from_a_second_thread(monkeypatch, built_lib):
    s = _solver(monkeypatch, resident=False, big=True)
    N = 1200                                         # (far fewer than the ~2,300 after which the recurrence residual underflows to zero)
    full = s.solve(1e-30, N)                         # unreachable within N: runs all N iterations
    assert full.iterations == N and not full.converged and not full.cancelled
    t_full = full.solve_ms
    canceller = threading.Timer(0.25 * t_full * 1e-3, lambda: capi.check(s.lib.avs_cancel(s.h)))
    canceller.start()
    t0 = time.perf_counter()
    info = s.solve(1e-30, N)                         # (ctypes releases the GIL: the timer thread runs while this call is inside the library)
    el = time.perf_counter() - t0
    canceller.join()
    assert info.cancelled == 1 and info.converged == 0
    assert 0 < info.iterations < N, info.iterations
    assert el < 0.9 * t_full * 1e-3 + 0.3
    assert np.all(np.isfinite(s.solution()))
    # the request is consumed: the next solve runs to convergence
    again = s.solve(1e-8, 20000)
    assert again.converged == 1 and again.cancelled == 0
    s.close()


def test_cancel_before_the_solve_cancels_the_next_one(monkeypatch, built_lib):
    for resident in (False, True):
        s = _solver(monkeypatch, resident)
        capi.check(s.lib.avs_cancel(s.h))
        info = s.solve(1e-8, 5000)
        assert info.cancelled == 1 and info.converged == 0 and info.iterations == 0
        x = s.solution()
        assert np.array_equal(x, s.initial_guess())      # nothing was iterated
        info = s.solve(1e-8, 5000)
        assert info.converged == 1 and info.cancelled == 0
        s.close()


def test_cancel_a_partitioned_solve(monkeypatch, built_lib):
    """world = 1 through the direct transport (the request travels with the CG sums: every rank leaves in the same round)"""
    import ctypes as C
    s = _solver(monkeypatch, resident=False, big=True)
    buf = (C.c_uint8 * capi.UNIQUE_ID_BYTES)()
    capi.check(s.lib.avs_dist_get_unique_id(buf))
    capi.check(s.lib.avs_dist_init(s.h, buf, 0, 1))
    s.dist_assemble()
    N = 1200
    full = s.dist_solve(1e-30, N)
    assert full.iterations == N and not full.cancelled
    canceller = threading.Timer(0.25 * full.solve_ms * 1e-3, lambda: capi.check(s.lib.avs_cancel(s.h)))
    canceller.start()
    info = s.dist_solve(1e-30, N)
    canceller.join()
    assert info.cancelled == 1 and info.converged == 0 and 0 < info.iterations < N
    again = s.dist_solve(1e-8, 20000)
    assert again.converged == 1 and again.cancelled == 0
    s.close()


def test_cancel_during_a_resident_solve_is_consumed(monkeypatch, built_lib):
    """round-5 advisor finding: the CU-resident loop (one cooperative launch) cannot be interrupted, but a request that arrives DURING
    the launch must end with it -- reported when the launch stopped at max_iterations unconverged, consumed either way -- and never
    hit the next solve on the context (the shim keeps its context across substeps)."""
    s = _solver(monkeypatch, resident=True)
    N = 1500
    full = s.solve(1e-30, N)
    assert full.resident == 1 and full.iterations == N and not full.cancelled
    canceller = threading.Timer(0.3 * full.solve_ms * 1e-3, lambda: capi.check(s.lib.avs_cancel(s.h)))
    canceller.start()
    info = s.solve(1e-30, N)
    canceller.join()
    assert info.resident == 1 and info.iterations == N       # the launch ran to its end ...
    assert info.cancelled == 1 and info.converged == 0       # ... and reports the request it consumed
    again = s.solve(1e-8, 20000)                             # consumed: the next solve is not cancelled
    assert again.converged == 1 and again.cancelled == 0 and again.iterations > 0
    # a request that arrives after a converged resident solve started is consumed with it, too
    canceller = threading.Timer(0.0, lambda: capi.check(s.lib.avs_cancel(s.h)))
    info = s.solve(1e-8, 20000)
    canceller.start()
    canceller.join()
    capi.check(s.lib.avs_cancel_clear(s.h))                  # what the shim does after joining its watcher thread
    again = s.solve(1e-8, 20000)
    assert again.converged == 1 and again.cancelled == 0 and again.iterations > 0
    s.close()
