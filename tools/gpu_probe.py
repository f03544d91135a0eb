"""Ad-hoc GPU probe: prepass + assemble + SpMV variant sweep + solve at a given size."""
import argparse
import json
import sys
import time
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from adaptiveviscositysolver_amd import DevicePrepass, ViscositySolve, scenes
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import prepass_torch as prepass  # test infrastructure

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=256)
ap.add_argument("--levels", type=int, default=4)
ap.add_argument("--tol", type=float, default=1e-3)
ap.add_argument("--variants", type=str, default="1,2,3,4")
ap.add_argument("--repeats", type=int, default=50)
ap.add_argument("--scene", type=str, default="beam")
ap.add_argument("--torch-prepass", action="store_true")
a = ap.parse_args()
dev = torch.device("cuda:0")
t0 = time.time()
if a.scene == "sheet":
    sc = scenes.thin_sheet(a.n, a.levels, thickness_cells=24, device=dev)
elif a.scene == "sphere":
    sc = scenes.sphere(a.n, a.levels, device=dev)
else:
    sc = scenes.fat_beam(a.n, a.levels, variable_viscosity=(a.scene == "varvisc"), device=dev)
torch.cuda.synchronize(); t1 = time.time()
if a.torch_prepass:
    pyr = prepass.build_pyramid(sc)
    torch.cuda.synchronize(); t2 = time.time()
    print(f"scene {t1-t0:.2f}s tensor prepass {t2-t1:.2f}s levels {pyr.levels} nv {pyr.n_velocity} ne {pyr.n_edge} nc {pyr.n_center} "
          f"peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB", flush=True)
    s = ViscositySolve(sc.res, sc.dx, sc.dt, pyr.levels, device=0, probe=True)
    s.set_pyramid(pyr)
    del pyr
else:
    pp = DevicePrepass(sc.res, sc.dx, sc.levels)
    pi = pp.run(sc.liquid, sc.solid)
    torch.cuda.synchronize(); t2 = time.time()
    print(f"scene {t1-t0:.2f}s device prepass {t2-t1:.2f}s (weights {pi.weights_ms:.1f} octree {pi.octree_ms:.1f} classify {pi.classify_ms:.1f} "
          f"numbering {pi.number_ms:.1f} ms) levels {pi.levels} nv {pi.n_velocity} ne {pi.n_edge} nc {pi.n_center}", flush=True)
    s = ViscositySolve(sc.res, sc.dx, sc.dt, pi.levels, device=0, probe=True)
    pp.apply(s)
    pp.close()
s.set_scene_fields(sc)
torch.cuda.empty_cache()
ai = s.assemble()
print(f"assemble: stencils {ai.stencil_ms:.2f} ms, guess {ai.guess_ms:.2f} ms, system {ai.system_ms:.2f} ms; n {ai.n_velocity} nnz {ai.nnz} raw {ai.raw_triplets}", flush=True)
n, nnz = ai.n_velocity, ai.nnz
bytes_spmv = 12 * nnz + 4 * (n + 1) + 16 * n
for v in [int(x) for x in a.variants.split(",")]:
    ms = s.bench_spmv(v, a.repeats)
    print(f"spmv variant {v}: {ms*1e3:.1f} us  {bytes_spmv/ms/1e6:.1f} GB/s  ({bytes_spmv/ms/1e6/8000*100:.1f}% of 8 TB/s)", flush=True)
info = s.solve(a.tol, 2500)
fmt = s.matrix_format()
print(f"matrix format: {fmt.bytes_per_nonzero} B/nnz, {fmt.value_table_size} distinct values, column bits {fmt.column_bits}", flush=True)
print(f"solve tol {a.tol}: iters {info.iterations} conv {info.converged} err {info.error:.3e} {info.solve_ms:.2f} ms -> {info.iterations/info.solve_ms*1e3:.1f} it/s", flush=True)
