"""SpMV of the 1024^3 thin sheet (BASELINE configs[4]): brick-structured form (64-bit streamed words) against the windowed-column stream."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from adaptiveviscositysolver_amd import DevicePrepass, ViscositySolve, scenes
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
sc = scenes.thin_sheet(n, 5, thickness_cells=32, device=dev)
pp = DevicePrepass(sc.res, sc.dx, sc.levels); pi = pp.run(sc.liquid, sc.solid)
s = ViscositySolve(sc.res, sc.dx, sc.dt, pi.levels, probe=True); pp.apply(s); s.set_scene_fields(sc); pp.close()
del sc; torch.cuda.empty_cache()
ai = s.assemble()
f = s.matrix_format()
print("rows", ai.n_velocity, "nnz", ai.nnz, "tiles", f.brick_tiles, "patterns", f.brick_patterns, "pattern rows", f.brick_pattern_rows, "brick bytes", f.brick_bytes,
      "column_bits", f.column_bits, "windows", f.column_windows, "table", f.value_table_size)
for r in range(2): print("default fused-dot SpMV us:", s.bench_spmv(100, 100) * 1e3)
for r in range(2): print("stream kernel (variant 61) fused-dot us:", s.bench_spmv(61, 100) * 1e3)
info = s.solve(1e-3, 4000)
print("solve:", info.iterations, "iterations", info.solve_ms, "ms")
