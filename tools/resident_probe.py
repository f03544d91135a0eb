"""Debug aid for the CU-resident PCG: a few scenes through both GPU loops, iteration counts / errors / times side by side."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from adaptiveviscositysolver_amd import DevicePrepass, ViscositySolve, capi, scenes
os.environ["AVS_CG_RESIDENT_VERBOSE"] = "1"
dev = torch.device("cuda:0")
CASES = {"beam64w": lambda: scenes.fat_beam(64, 3, wall=True, device=dev), "sphere64": lambda: scenes.sphere(64, 4, device=dev),
         "sphere32": lambda: scenes.sphere(32, 3, device=dev),
         "beam128": lambda: scenes.fat_beam(128, 3, device=dev), "beam128L4": lambda: scenes.fat_beam(128, 4, device=dev),
         "beam256L5": lambda: scenes.fat_beam(256, 5, device=dev), "beam256L4": lambda: scenes.fat_beam(256, 4, device=dev),
         "beam320L4": lambda: scenes.fat_beam(320, 4, device=dev), "beam384L4": lambda: scenes.fat_beam(384, 4, device=dev), "hipbeam": lambda: scenes.viscous_beam_scene(device=dev),
         "hipbuckling": lambda: scenes.viscous_buckling_scene(device=dev)}
for name in (sys.argv[1:] or list(CASES)):
    sc = CASES[name]()
    fsc = scenes.crop_to_field(sc)
    pp = DevicePrepass(sc.res, sc.dx, sc.levels, field_res=sc.field_res)
    pi = pp.run(fsc.liquid, fsc.solid)
    s = ViscositySolve(sc.res, sc.dx, sc.dt, pi.levels, field_res=sc.field_res)
    pp.apply(s); s.set_scene_fields(fsc); pp.close()
    ai = s.assemble()
    rp, col, val, rhs = s.csr()
    L = np.diff(rp)
    print(name, "n", ai.n_velocity, "nnz", ai.nnz, "max row", int(L.max()), "rows>64", int((L > 64).sum()), flush=True)
    for tol in (1e-3, 1e-10):
        out = {}
        for res in ("0", "1"):
            s.set_solver_option(capi.OPTION_RESIDENT_LOOP, int(res))
            s.solve(tol, 5000)
            info = s.solve(tol, 5000)
            out[res] = (info.iterations, info.converged, info.error, info.solve_ms, info.resident, s.solution())
        a, b = out["0"], out["1"]
        rel = np.linalg.norm(a[5] - b[5]) / np.linalg.norm(a[5])
        print(f"  tol {tol:g}: standard it {a[0]} conv {a[1]} err {a[2]:.2e} {a[3]:.2f} ms ({a[0] / a[3]:.1f} it/ms) | resident[{b[4]}] it {b[0]} conv {b[1]} err {b[2]:.2e} "
              f"{b[3]:.2f} ms ({b[0] / max(b[3], 1e-9):.1f} it/ms) | rel diff {rel:.2e}", flush=True)
    s.close()
