"""Per 512-row SpMV tile of the brick-major system: entries whose column lies outside the tile's own rows, how many DISTINCT
columns that is (what a tile-local column cache would have to fetch) and how many 128-B lines of x those touch."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from adaptiveviscositysolver_amd import DevicePrepass, ViscositySolve, capi, scenes
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=512); ap.add_argument("--levels", type=int, default=4); ap.add_argument("--tile", type=int, default=512)
a = ap.parse_args()
dev = torch.device("cuda:0")
sc = scenes.fat_beam(a.n, a.levels, device=dev)
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
row_new = inv[torch.repeat_interleave(torch.arange(n, device=dev), lens)]
col_new = inv[col.long()]; del col
out = {"rows": n, "nnz": nnz}
for T in (a.tile, 2 * a.tile, 4 * a.tile):
    tile = row_new // T
    ntiles = int((n + T - 1) // T)
    outside = (col_new // T) != tile
    t_o, c_o = tile[outside], col_new[outside]
    uk = torch.unique(t_o * n + c_o)                     # distinct (tile, column)
    ut = uk // n
    ul = torch.unique(t_o * n + (c_o // 16) * 16) // n   # distinct (tile, line)
    cnt = lambda idx: torch.zeros(ntiles, dtype=torch.int64, device=dev).index_add_(0, idx, torch.ones_like(idx))
    e, d, l = cnt(t_o), cnt(ut), cnt(ul)
    q = lambda v: [float(x) for x in torch.quantile(v.double(), torch.tensor([0.1, 0.5, 0.9, 0.99, 1.0], device=dev, dtype=torch.float64))]
    out[f"tile_{T}"] = {"tiles": ntiles, "outside_entries_share": float(outside.double().mean()), "outside_entries_per_tile": q(e),
                        "distinct_outside_columns_per_tile": q(d), "distinct_lines_per_tile": q(l),
                        "remote_list_bytes_over_word_bytes": float(d.sum() * 4) / (nnz * 4)}
print(json.dumps(out, indent=1))
