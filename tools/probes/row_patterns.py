"""How repetitive are the rows of the brick-major system?  A row's PATTERN = its sequence of (column - row, value code).
Reports: distinct patterns in the whole matrix, coverage by the most frequent ones, and -- per 512-row tile (the SpMV
workgroup) -- the words a tile-local pattern dictionary would have to store next to one id per row."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from adaptiveviscositysolver_amd import DevicePrepass, ViscositySolve, capi, scenes

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=512)
ap.add_argument("--levels", type=int, default=4)
ap.add_argument("--scene", default="beam")
ap.add_argument("--tile", type=int, default=512)
a = ap.parse_args()
dev = torch.device("cuda:0")
if a.scene == "beam": sc = scenes.fat_beam(a.n, a.levels, device=dev)
elif a.scene == "hipbeam": sc = scenes.viscous_beam_scene(dev, 1)
elif a.scene == "hipbuckling": sc = scenes.viscous_buckling_scene(dev, 1)
else: sc = scenes.thin_sheet(a.n, a.levels, thickness_cells=32, device=dev)
fres = getattr(sc, "field_res", None)
pp = DevicePrepass(sc.res, sc.dx, sc.levels, field_res=fres)
if fres is not None: sc = scenes.crop_to_field(sc)
pi = pp.run(sc.liquid, sc.solid)
s = ViscositySolve(sc.res, sc.dx, sc.dt, pi.levels, field_res=fres)
pp.apply(s); s.set_scene_fields(sc); pp.close()
nx, ny, nz = sc.res
del sc; torch.cuda.empty_cache()
ai = s.assemble()
n, nnz = int(ai.n_velocity), int(ai.nnz)
rp = torch.empty(n + 1, dtype=torch.int32, device=dev); col = torch.empty(nnz, dtype=torch.int32, device=dev); val = torch.empty(nnz, dtype=torch.float64, device=dev)
capi.check(s.lib.avs_get_csr(s.h, rp.data_ptr(), col.data_ptr(), val.data_ptr(), None, capi.MEM_DEVICE))
tab = torch.empty((n, 4), dtype=torch.int32, device=dev)
capi.check(s.lib.avs_get_dof_table(s.h, capi.INDEX_VELOCITY, tab.data_ptr(), capi.MEM_DEVICE))
s.close()
lv = (tab[:, 0] & 0xff).long()
P = [(tab[:, 1 + k].long() << lv) for k in range(3)]
P = [P[0].clamp(max=nx - 1), P[1].clamp(max=ny - 1), P[2].clamp(max=nz - 1)]
nbx, nby = (nx + 7) >> 3, (ny + 7) >> 3
key = ((P[2] >> 3) * nby + (P[1] >> 3)) * nbx + (P[0] >> 3)
key = (key << 9) | ((P[2] & 7) << 6) | ((P[1] & 7) << 3) | (P[0] & 7)
perm = torch.sort(key, stable=True).indices
inv = torch.empty_like(perm); inv[perm] = torch.arange(n, device=dev)
del tab, P, key
lens = (rp[1:] - rp[:-1]).long()
rows_old = torch.repeat_interleave(torch.arange(n, device=dev), lens)
row_new = inv[rows_old]
j_in_row = torch.arange(nnz, device=dev) - rp[:-1].long()[rows_old]
col_new = inv[col.long()]
del rows_old, col
uv, code = torch.unique(val, return_inverse=True)
del val
delta = col_new - row_new
M1, M2, M3 = -7046029254386353131, -4417276706812531889, 1609587929392839161
h = ((delta * M1) ^ (code * M2)) * M3
h = (h ^ (h >> 29)) * (2 * j_in_row + 1) * M1
rowh = torch.zeros(n, dtype=torch.int64, device=dev).index_add_(0, row_new, h)
lens_new = lens[perm]
rowh = rowh * 31 + lens_new
out = {"scene": a.scene, "rows": n, "nnz": nnz, "values": int(uv.numel())}
u, cnt = torch.unique(rowh, return_counts=True)
order = torch.argsort(cnt, descending=True)
# nnz covered by the top-K global patterns
first = torch.zeros_like(u); 
srt, idx = torch.sort(rowh); 
plen = torch.zeros(u.numel(), dtype=torch.int64, device=dev)
pos = torch.searchsorted(srt, u)
plen = lens_new[idx[pos]]
cov = {}
cs = torch.cumsum((cnt * plen)[order], 0)
for K in (256, 4096, 65536, 1 << 20):
    k = min(K, u.numel()); cov[str(K)] = float(cs[k - 1]) / nnz
out["global"] = {"distinct_patterns": int(u.numel()), "nnz_coverage_by_top": cov, "dictionary_words_all": int(plen.sum())}
for T in (a.tile, 4 * a.tile, 16 * a.tile):
    tile = torch.arange(n, device=dev) // T
    ntiles = int(tile[-1]) + 1
    # distinct (tile, pattern)
    k2 = torch.stack([tile, rowh], 1)
    uk, inv2, c2 = torch.unique(k2, dim=0, return_inverse=True, return_counts=True)
    # words of each distinct (tile, pattern) = its length
    firstrow = torch.zeros(uk.shape[0], dtype=torch.int64, device=dev).scatter_(0, inv2, torch.arange(n, device=dev))
    w = lens_new[firstrow]
    per_tile = torch.zeros(ntiles, dtype=torch.int64, device=dev).index_add_(0, uk[:, 0], torch.ones_like(w))
    words_tile = torch.zeros(ntiles, dtype=torch.int64, device=dev).index_add_(0, uk[:, 0], w)
    nnz_tile = torch.zeros(ntiles, dtype=torch.int64, device=dev).index_add_(0, tile, lens_new)
    # a tile uses the dictionary only when it pays: words + 2 B per row < explicit words
    dict_bytes = words_tile * 4 + 2 * T
    expl_bytes = nnz_tile * 4
    best = torch.minimum(dict_bytes, expl_bytes)
    q = torch.quantile(per_tile.double(), torch.tensor([0.1, 0.5, 0.9], device=dev, dtype=torch.float64)).tolist()
    out[f"tile_{T}"] = {"tiles": ntiles, "distinct_per_tile_p10_p50_p90": q, "dictionary_words": int(words_tile.sum()),
                        "words_ratio": float(words_tile.sum()) / nnz, "bytes_ratio_best_of_both": float(best.sum()) / float(expl_bytes.sum()),
                        "tiles_using_dictionary": float((dict_bytes < expl_bytes).double().mean())}
# structure only (column deltas), and codes only
def row_hash(x):
    hh = (x * M1) * M3
    hh = (hh ^ (hh >> 29)) * (2 * j_in_row + 1) * M1
    return torch.zeros(n, dtype=torch.int64, device=dev).index_add_(0, row_new, hh) * 31 + lens_new
for name, src in (("structure", delta), ("codes", code)):
    rh = row_hash(src)
    res = {"distinct_global": int(torch.unique(rh).numel())}
    for T in (a.tile, 4 * a.tile, 16 * a.tile):
        tile = torch.arange(n, device=dev) // T
        uk, inv2 = torch.unique(torch.stack([tile, rh], 1), dim=0, return_inverse=True)
        firstrow = torch.zeros(uk.shape[0], dtype=torch.int64, device=dev).scatter_(0, inv2, torch.arange(n, device=dev))
        w = lens_new[firstrow]
        ntiles = int(tile[-1]) + 1
        per_tile = torch.zeros(ntiles, dtype=torch.int64, device=dev).index_add_(0, uk[:, 0], torch.ones_like(w))
        words_tile = torch.zeros(ntiles, dtype=torch.int64, device=dev).index_add_(0, uk[:, 0], w)
        q = torch.quantile(per_tile.double(), torch.tensor([0.1, 0.5, 0.9, 1.0], device=dev, dtype=torch.float64)).tolist()
        qw = torch.quantile(words_tile.double(), torch.tensor([0.5, 0.9, 1.0], device=dev, dtype=torch.float64)).tolist()
        res[f"tile_{T}"] = {"distinct_per_tile_p10_p50_p90_max": q, "dictionary_entries": int(words_tile.sum()), "entries_ratio": float(words_tile.sum()) / nnz,
                            "dictionary_entries_per_tile_p50_p90_max": qw}
    out[name] = res
# how far do the deltas reach (bits of a tile-relative / row-relative column)?
ad = delta.abs()
out["delta_abs_quantiles"] = {str(qq): float(torch.quantile(ad[::64].double(), qq)) for qq in (0.5, 0.9, 0.99, 0.999)}
out["delta_abs_max"] = int(ad.max())
print(json.dumps(out, indent=1))
