"""Soak of the CU-resident loop: the same systems solved many times; every solve must converge in the same number of iterations."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from adaptiveviscositysolver_amd import DevicePrepass, ViscositySolve, scenes
dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for name, make in (("beam128", lambda: scenes.fat_beam(128, 3, device=dev)), ("hipbuckling", lambda: scenes.viscous_buckling_scene(device=dev)),
                   ("beam256 (streamed rows)", lambda: scenes.fat_beam(256, 4, device=dev))):
    sc = make(); fsc = scenes.crop_to_field(sc)
    pp = DevicePrepass(sc.res, sc.dx, sc.levels, field_res=sc.field_res); pi = pp.run(fsc.liquid, fsc.solid)
    s = ViscositySolve(sc.res, sc.dx, sc.dt, pi.levels, field_res=sc.field_res); pp.apply(s); s.set_scene_fields(fsc); pp.close(); s.assemble()
    first = s.solve(1e-6, 5000)
    t0 = time.time(); bad = 0
    for i in range(reps):
        info = s.solve(1e-6, 5000)
        if info.resident != 1 or info.converged != 1 or info.iterations != first.iterations:
            bad += 1
            print("  deviation at", i, info.resident, info.converged, info.iterations, first.iterations)
    print(f"{name}: {reps} solves, {first.iterations} iterations each, {bad} deviations, {(time.time() - t0) / reps * 1e3:.2f} ms per solve", flush=True)
    s.close()
