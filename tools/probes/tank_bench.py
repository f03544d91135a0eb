"""In-loop rate of one scene through the product library (AVS_LIB_PATH selects the build): python tools/probes/tank_bench.py tank 512"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from adaptiveviscositysolver_amd import DevicePrepass, ViscositySolve, scenes, capi
dev = torch.device("cuda:0")
kind, n = sys.argv[1], int(sys.argv[2])
sc = {"tank": lambda: scenes.tank(n, 4, device=dev), "beam": lambda: scenes.fat_beam(n, 4, device=dev),
      "beam_mu": lambda: scenes.fat_beam(n, 4, variable_viscosity=True, device=dev),
      "sheet": lambda: scenes.thin_sheet(n, 4 if n <= 512 else 5, thickness_cells=32, device=dev)}[kind]()
pp = DevicePrepass(sc.res, sc.dx, sc.levels); pi = pp.run(sc.liquid, sc.solid)
s = ViscositySolve(sc.res, sc.dx, sc.dt, pi.levels); pp.apply(s); s.set_scene_fields(sc); pp.close()
s.set_solver_option(capi.OPTION_RESIDENT_LOOP, 0)
s.assemble()
s.solve(1e-3, 3000)
best = (0, 0, 0)
for _ in range(3):
    info = s.solve(1e-3, 3000)
    r = info.iterations / info.solve_ms * 1e3
    if r > best[0]: best = (r, info.spmv_ms * 1e3, info.iterations)
print(f"{kind} {n} lib={os.path.basename(capi.LIB_PATH)}: {best[0]:.0f} it/s, SpMV {best[1]:.1f} us, {best[2]} iterations, tiles {s.matrix_format().brick_tiles}", flush=True)
