"""BASELINE configs[4] experiment: is a blocked / ELL view of the octree viscosity matrix worth it on MI355X?

Builds the brick-major (8^3) system exactly as the solver numbers it, converts it to SELL-C-sigma (C = 64: one wavefront per
slice, sigma-window sort by row length) with torch tensor ops, and reports: padding ratio, time of the SELL kernel
(avs_spmv_sell), time of the library's kernels on the same matrix (plain 12-B tile kernel, default compressed kernel), and
the fill of b x b blocks (would a blocked view give MFMA a dense tile contraction?)."""
import argparse
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from adaptiveviscositysolver_amd import DevicePrepass, ViscositySolve, capi, scenes

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="sheet", choices=("sheet", "beam"))
ap.add_argument("--n", type=int, default=1024)
ap.add_argument("--levels", type=int, default=5)
ap.add_argument("--sigma", type=int, default=512)
ap.add_argument("--repeats", type=int, default=50)
a = ap.parse_args()
dev = torch.device("cuda:0")
sc = scenes.thin_sheet(a.n, a.levels, thickness_cells=32, device=dev) if a.scene == "sheet" else scenes.fat_beam(a.n, a.levels, device=dev)
pp = DevicePrepass(sc.res, sc.dx, sc.levels)
pi = pp.run(sc.liquid, sc.solid)
s = ViscositySolve(sc.res, sc.dx, sc.dt, pi.levels, probe=True)
pp.apply(s)
s.set_scene_fields(sc)
pp.close()
del sc
torch.cuda.empty_cache()
ai = s.assemble()
n, nnz = int(ai.n_velocity), int(ai.nnz)
out = {"scene": a.scene, "n": a.n, "levels": int(pi.levels), "rows": n, "nnz": nnz}
lib = s.lib
rp = torch.empty(n + 1, dtype=torch.int32, device=dev)
col = torch.empty(nnz, dtype=torch.int32, device=dev)
val = torch.empty(nnz, dtype=torch.float64, device=dev)
capi.check(lib.avs_get_csr(s.h, rp.data_ptr(), col.data_ptr(), val.data_ptr(), None, capi.MEM_DEVICE))
# library kernels on the brick-major system (plain 12-B tile kernel = variant 24, default = compressed form)
out["library_us"] = {"plain_12B_tile": s.bench_spmv(14, a.repeats) * 1e3, "default": s.bench_spmv(0, a.repeats) * 1e3}
out["matrix_format"] = {"bytes_per_nonzero": s.matrix_format().bytes_per_nonzero, "table": s.matrix_format().value_table_size}

# brick-major permutation, as avs_reorder.hip builds it (stable sort of 8^3 brick keys)
tab = torch.empty((n, 4), dtype=torch.int32, device=dev)
capi.check(lib.avs_get_dof_table(s.h, capi.INDEX_VELOCITY, tab.data_ptr(), capi.MEM_DEVICE))
s.close()
lv = (tab[:, 0] & 0xff).long()
p3 = [(tab[:, 1 + k].long() << lv).clamp(max=a.n - 1) >> 3 for k in range(3)]
nb = (a.n + 7) >> 3
key = (p3[2] * nb + p3[1]) * nb + p3[0]
perm = torch.sort(key, stable=True).indices            # new -> old
inv = torch.empty_like(perm)
inv[perm] = torch.arange(n, device=dev)
del tab, lv, p3, key
lens = (rp[1:] - rp[:-1]).long()
lens_new = lens[perm]
# rows of the new numbering, entries in the old in-row order
rows_old = torch.repeat_interleave(torch.arange(n, device=dev), lens)
row_new = inv[rows_old]
j_in_row = torch.arange(nnz, device=dev) - rp[:-1].long()[rows_old]
col_new = inv[col.long()]
del rows_old, col
# block fill of the brick-major matrix
fill = {}
for b in (4, 8, 16):
    k2 = (row_new // b) * ((n + b - 1) // b) + (col_new // b)
    nblocks = int(torch.unique(k2).numel())
    fill[str(b)] = {"blocks": nblocks, "fill": nnz / (nblocks * b * b)}
    del k2
out["block_fill"] = fill
# SELL-64-sigma: sort rows by length (descending) inside windows of sigma rows
sig = a.sigma
win = torch.arange(n, device=dev) // sig
order = torch.sort(win * 1024 + (1023 - lens_new.clamp(max=1023)), stable=True).indices   # position -> new row
pos_of_row = torch.empty_like(order)
pos_of_row[order] = torch.arange(n, device=dev)
nsl = (n + 63) // 64
lens_sorted = torch.zeros(nsl * 64, dtype=torch.long, device=dev)
lens_sorted[:n] = lens_new[order]
width = lens_sorted.view(nsl, 64).max(dim=1).values
slice_ptr = torch.zeros(nsl + 1, dtype=torch.int64, device=dev)
slice_ptr[1:] = torch.cumsum(width * 64, 0)
padded = int(slice_ptr[-1].item())
out["sell"] = {"C": 64, "sigma": sig, "slices": nsl, "padded_entries": padded, "padding_ratio": padded / nnz,
               "max_row": int(lens.max().item()), "mean_row": nnz / n}
scol = torch.zeros(padded, dtype=torch.int32, device=dev)
sval = torch.zeros(padded, dtype=torch.float64, device=dev)
p = pos_of_row[row_new]
dst = slice_ptr[p // 64] + j_in_row * 64 + (p % 64)
scol[dst] = col_new.int()
sval[dst] = val
del dst, p, row_new, j_in_row, col_new, val
x = torch.randn(n, dtype=torch.float64, device=dev)
y = torch.empty(nsl * 64, dtype=torch.float64, device=dev)
ms = C.c_double()
capi.check(lib.avs_spmv_sell(nsl, slice_ptr.data_ptr(), scol.data_ptr(), sval.data_ptr(), x.data_ptr(), y.data_ptr(), a.repeats, None,
                             C.byref(ms)))
torch.cuda.synchronize()
alg = 12 * nnz + 4 * (n + 1) + 16 * n
out["sell"]["us"] = ms.value * 1e3
out["sell"]["frac_8d"] = alg / (ms.value * 1e-3) / 1e9 / 8000.0
out["sell"]["streamed_bytes"] = 12 * padded + 8 * (nsl + 1) + 16 * n
for k, v in out["library_us"].items():
    out.setdefault("library_frac_8d", {})[k] = alg / (v * 1e-6) / 1e9 / 8000.0
print(json.dumps(out, indent=1))
