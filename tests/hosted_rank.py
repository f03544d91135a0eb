"""One rank of a HOSTED multi-process solve (tests/test_gpu_dist.py::test_processes_direct_transport).

Real one-process-per-rank execution of the direct transport on a 1-GPU box: both processes use cuda:0, map each other's
comm block through HIP IPC handles and run the flag-based halo exchange / all-gather between two processes.  The blobs
travel through files in `workdir` (the "host program" of a hosted group may be anything: MPI, gloo, files)."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def wait_for(path, timeout=120.0):
    t0 = time.time()
    while not os.path.exists(path):
        if time.time() - t0 > timeout:
            raise TimeoutError(path)
        time.sleep(0.02)


def main():
    workdir, rank, world, scene, tol = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], float(sys.argv[5])
    import torch
    from adaptiveviscositysolver_amd import DevicePrepass, ViscositySolve, capi, scenes
    dev = torch.device("cuda:0")
    sc = {"beam": lambda: scenes.fat_beam(64, 3, device=dev),
          "varvisc": lambda: scenes.fat_beam(64, 3, variable_viscosity=True, device=dev),
          "beam128": lambda: scenes.fat_beam(128, 3, device=dev),
          "beam128_brick": lambda: scenes.fat_beam(128, 3, device=dev),        # (AVS_BRICK=1 in the environment: the brick-structured form)
          "beam128L4_brick": lambda: scenes.fat_beam(128, 4, device=dev),
          "varvisc128_brick": lambda: scenes.fat_beam(128, 4, variable_viscosity=True, device=dev),
          "varvisc128": lambda: scenes.fat_beam(128, 4, variable_viscosity=True, device=dev)}[scene]()
    pp = DevicePrepass(sc.res, sc.dx, sc.levels)
    pi = pp.run(sc.liquid, sc.solid)
    # the stale-entry injection hook exists in the probe build only
    s = ViscositySolve(sc.res, sc.dx, sc.dt, pi.levels, probe=bool(os.environ.get("AVS_DIST_INJECT_STALE")))
    pp.apply(s)
    s.set_scene_fields(sc)
    capi.check(s.lib.avs_dist_init_hosted(s.h, rank, world))
    s.dist_assemble()
    blob = (C.c_uint8 * capi.DIST_BLOB_BYTES)()
    capi.check(s.lib.avs_dist_export_blob(s.h, blob))
    tmp = os.path.join(workdir, f"blob_{rank}.tmp")
    open(tmp, "wb").write(bytes(blob))
    os.rename(tmp, os.path.join(workdir, f"blob_{rank}.bin"))
    allb = b""
    for q in range(world):
        wait_for(os.path.join(workdir, f"blob_{q}.bin"))
        allb += open(os.path.join(workdir, f"blob_{q}.bin"), "rb").read()
    buf = (C.c_uint8 * len(allb)).from_buffer_copy(allb)
    runs = []
    resident = 0
    try:
        capi.check(s.lib.avs_dist_import_blobs(s.h, buf))    # connects the comm blocks and runs the transport self-test
        for _ in range(2):                                   # twice: the second solve replays the captured graph
            info = s.dist_solve(tol, 5000)
            runs.append((info.iterations, info.converged, info.error))
            resident = int(info.resident)
    except capi.AvsError as e:                               # (the stale-halo tests expect exactly this)
        open(os.path.join(workdir, f"err_{rank}.txt"), "w").write(f"{e.status}\n{e}")
        open(os.path.join(workdir, f"done_{rank}"), "w").write("failed")
        sys.stderr.write(str(e))
        sys.exit(7)
    x = s.dist_solution()                                 # hosted group: owned entries, zeros elsewhere
    ci = s.dist_comm_info()
    np.save(os.path.join(workdir, f"x_{rank}.npy"), x)
    np.save(os.path.join(workdir, f"info_{rank}.npy"), np.array([runs[0][0], runs[0][1], runs[1][0], runs[1][1],
                                                                  s.plan_sizes.n_own, s.plan_sizes.n_halo,
                                                                  1 if ci["transport"] == "direct" else 0,
                                                                  ci["rccl_calls_per_iteration"], ci["launches_per_iteration"],
                                                                  s.matrix_format().tile_local_tables, s.matrix_format().column_windows,
                                                                  resident, ci["selftest_rounds"], ci["selftest_bad_entries"], 1 if ci["paranoid"] else 0],
                                                                 np.float64))
    np.save(os.path.join(workdir, f"fmt_{rank}.npy"), np.array([s.matrix_format().brick_tiles, s.matrix_format().brick_pattern_rows], np.float64))
    # keep the comm block alive until every rank has finished (a peer may still be reading its own copy of the flags)
    open(os.path.join(workdir, f"done_{rank}"), "w").write("ok")
    for q in range(world):
        wait_for(os.path.join(workdir, f"done_{q}"))
    s.close()


if __name__ == "__main__":
    main()
