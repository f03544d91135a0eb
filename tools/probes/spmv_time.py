"""Default SpMV of the 512^3 beam, timed by avs_bench_spmv (which also checks y against the plain CSR kernel bit for bit)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from adaptiveviscositysolver_amd import DevicePrepass, ViscositySolve, scenes
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
kind = os.environ.get("SPMV_SCENE", "beam")   # beam | sheet (thickness SPMV_SHEET_CELLS, default 32) | tank
sc = (scenes.thin_sheet(n, 4 if n <= 512 else 5, thickness_cells=int(os.environ.get("SPMV_SHEET_CELLS", "32")), device=dev) if kind == "sheet"
      else scenes.tank(n, 4, device=dev) if kind == "tank" else scenes.fat_beam(n, 4, device=dev))
pp = DevicePrepass(sc.res, sc.dx, sc.levels); pi = pp.run(sc.liquid, sc.solid)
s = ViscositySolve(sc.res, sc.dx, sc.dt, pi.levels, probe=True); pp.apply(s); s.set_scene_fields(sc); pp.close(); s.assemble()
reps = int(os.environ.get("SPMV_REPEATS", "200"))
for r in range(3): print("default SpMV us:", s.bench_spmv(0, reps) * 1e3)
for r in range(3): print("fused-dot SpMV us:", s.bench_spmv(100, reps) * 1e3)
for r in range(2): print("stream kernel (variant 61) fused-dot us:", s.bench_spmv(61, reps) * 1e3)
print(s.matrix_format().brick_tiles, "tiles")
