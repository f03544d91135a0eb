"""SELL-64 view of the brick-major system (rows sorted by length inside each 512-row SpMV tile, slices of 64 rows, slot j = j-th
entry of every row): how many (slice, slot) pairs contain at least one column outside the tile's LDS window (= a global gather
instruction for the wave), and how full those instructions would be.  Compared with the stream kernel's 8 slots x 2 passes."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from adaptiveviscositysolver_amd import DevicePrepass, ViscositySolve, capi, scenes
dev = torch.device("cuda:0")
n0 = int(sys.argv[1]) if len(sys.argv) > 1 else 512
sc = scenes.fat_beam(n0, 4, device=dev)
pp = DevicePrepass(sc.res, sc.dx, sc.levels); pi = pp.run(sc.liquid, sc.solid)
s = ViscositySolve(sc.res, sc.dx, sc.dt, pi.levels); pp.apply(s); s.set_scene_fields(sc); pp.close()
nx, ny, nz = sc.res
del sc; torch.cuda.empty_cache()
ai = s.assemble(); n, nnz = int(ai.n_velocity), int(ai.nnz)
rp = torch.empty(n + 1, dtype=torch.int32, device=dev); col = torch.empty(nnz, dtype=torch.int32, device=dev); val = torch.empty(nnz, dtype=torch.float64, device=dev)
capi.check(s.lib.avs_get_csr(s.h, rp.data_ptr(), col.data_ptr(), val.data_ptr(), None, capi.MEM_DEVICE)); del val
tab = torch.empty((n, 4), dtype=torch.int32, device=dev)
capi.check(s.lib.avs_get_dof_table(s.h, capi.INDEX_VELOCITY, tab.data_ptr(), capi.MEM_DEVICE)); s.close()
lv = (tab[:, 0] & 0xff).long()
P = [(tab[:, 1 + k].long() << lv) for k in range(3)]
P = [P[0].clamp(max=nx - 1), P[1].clamp(max=ny - 1), P[2].clamp(max=nz - 1)]
nbx, nby = (nx + 7) >> 3, (ny + 7) >> 3
key = ((((P[2] >> 3) * nby + (P[1] >> 3)) * nbx + (P[0] >> 3)) << 9) | ((P[2] & 7) << 6) | ((P[1] & 7) << 3) | (P[0] & 7)
perm = torch.sort(key, stable=True).indices
inv = torch.empty_like(perm); inv[perm] = torch.arange(n, device=dev)
del tab, P, key
lens = (rp[1:] - rp[:-1]).long()
rows_old = torch.repeat_interleave(torch.arange(n, device=dev), lens)
row_new = inv[rows_old]
j_in_row = torch.arange(nnz, device=dev) - rp[:-1].long()[rows_old]
col_new = inv[col.long()]; del col, rows_old
lens_new = lens[perm]
T = 512
tile = row_new // T
outside = (col_new // T) != tile
out = {"rows": n, "nnz": nnz}
for sort_rows in (False, True):
    if sort_rows:   # rows sorted by length (descending, stable) inside each tile
        k2 = (torch.arange(n, device=dev) // T) * 4096 + (4095 - lens_new.clamp(max=4095))
        order = torch.sort(k2, stable=True).indices          # position -> row
        pos = torch.empty_like(order); pos[order] = torch.arange(n, device=dev)
    else:
        pos = torch.arange(n, device=dev)
    slice_of_row = pos // 64
    nsl = int(slice_of_row.max()) + 1
    slice_len = torch.zeros(nsl, dtype=torch.int64, device=dev).scatter_reduce_(0, slice_of_row, lens_new, "amax")
    padded = int(slice_len.sum()) * 64
    sl_e = slice_of_row[row_new]
    keyslot = sl_e * 64 + j_in_row.clamp(max=63)
    any_out = torch.zeros(nsl * 64, dtype=torch.int64, device=dev).index_add_(0, keyslot, outside.long())
    slots_total = int(slice_len.sum())
    slots_with = int((any_out > 0).sum())
    lanes = float(any_out[any_out > 0].double().mean())
    out["sorted" if sort_rows else "natural"] = {"padding": padded / nnz, "slots_per_slice_mean": slots_total / nsl, "slots_with_a_global_gather": slots_with / slots_total,
                                                 "gather_instructions_per_512_rows": slots_with / (n / 512), "active_lanes_per_gather": lanes}
out["stream_kernel_gather_instructions_per_512_rows"] = 8 * 2 * 8
print(json.dumps(out, indent=1))
