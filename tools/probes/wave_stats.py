"""Round 6: load balance of the brick kernel's row walk and the fill runs' lengths (avs_brick_wave_stats, probe build): python tools/probes/wave_stats.py [512] ; SPMV_SCENE=beam|sheet|tank"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from adaptiveviscositysolver_amd import DevicePrepass, ViscositySolve, scenes, capi
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
kind = os.environ.get("SPMV_SCENE", "beam")
sc = (scenes.thin_sheet(n, 4 if n <= 512 else 5, thickness_cells=32, device=dev) if kind == "sheet" else scenes.tank(n, 4, device=dev) if kind == "tank"
      else scenes.fat_beam(n, 4, variable_viscosity=True, device=dev) if kind == "varvisc" else scenes.fat_beam(n, 4, device=dev))
pp = DevicePrepass(sc.res, sc.dx, sc.levels); pi = pp.run(sc.liquid, sc.solid)
s = ViscositySolve(sc.res, sc.dx, sc.dt, pi.levels, probe=True); pp.apply(s); s.set_scene_fields(sc); pp.close(); s.assemble()
print(kind, n, s.matrix_format().brick_tiles, "tiles", flush=True)
out = (C.c_double * 6)()
capi.check(s.lib.avs_brick_wave_stats(s.h, out))
