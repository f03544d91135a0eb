"""Variable-viscosity SpMV probe: assembles the fat beam with mu(x) = 200 (1 + 9x) and times the tile-table kernel
geometries (variants 51-56), their fused-dot forms (151-156) and the library default, per launch (HIP events)."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from adaptiveviscositysolver_amd import DevicePrepass, ViscositySolve, scenes

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=512)
ap.add_argument("--levels", type=int, default=4)
ap.add_argument("--repeats", type=int, default=50)
ap.add_argument("--variants", default="0,100,51,52,53,54,151,152,153,154")
a = ap.parse_args()
dev = torch.device("cuda:0")
sc = scenes.fat_beam(a.n, a.levels, variable_viscosity=True, device=dev)
pp = DevicePrepass(sc.res, sc.dx, sc.levels)
pi = pp.run(sc.liquid, sc.solid)
s = ViscositySolve(sc.res, sc.dx, sc.dt, pi.levels, probe=True)
pp.apply(s)
s.set_scene_fields(sc)
pp.close()
ai = s.assemble()
fmt = s.matrix_format()
n, nnz = ai.n_velocity, ai.nnz
alg = 12 * nnz + 4 * (n + 1) + 16 * n
out = {"n": n, "nnz": nnz, "bytes_per_nonzero": fmt.bytes_per_nonzero, "tile_local_tables": fmt.tile_local_tables,
       "table_entries": fmt.value_table_size, "table_bytes_per_nonzero": 8.0 * fmt.value_table_size / nnz, "variants": {}}
for v in [int(x) for x in a.variants.split(",")]:
    try:
        ms = s.bench_spmv(v, a.repeats)
        out["variants"][v] = {"us": ms * 1e3, "frac_8d": alg / (ms * 1e-3) / 1e9 / 8000.0}
    except Exception as e:  # noqa: BLE001
        out["variants"][v] = {"error": str(e)[:200]}
info = s.solve(1e-3, 2500)
out["solve"] = {"iterations": info.iterations, "ms": info.solve_ms, "spmv_us": info.spmv_ms * 1e3}
print(json.dumps(out, indent=1))
