"""Brick-structured SpMV format, prototype statistics (round 4, review item 1).

Tile = the rows of one 8^3 brick of fine cells (brick-major, cell-major numbering of avs_reorder.hip).  Every column a tile reads gets a
tile-LOCAL slot: level-0 faces at brick-relative cell (-1..8)^3 -> the padded 10x10x10x3 lattice (translation invariant), any other
column -> an "extra" slot.  A row's PATTERN = its sequence of (slot - own slot, value code) in the stored column order.  Reports how
many rows share patterns inside a tile, how many runs the LDS fill needs, and the bytes the form would stream.

  --source oracle : CPU oracle (no GPU; <= 256^3)        --source device : product path on cuda:0 (512^3)
"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=128)
ap.add_argument("--levels", type=int, default=4)
ap.add_argument("--scene", default="beam")
ap.add_argument("--source", default="oracle")
ap.add_argument("--out", default="")
a = ap.parse_args()
torch.set_num_threads(8)

from adaptiveviscositysolver_amd import scenes


def make_scene(dev):
    if a.scene == "beam": return scenes.fat_beam(a.n, a.levels, device=dev)
    if a.scene == "sphere": return scenes.sphere(a.n, a.levels, device=dev)
    if a.scene == "varvisc": return scenes.fat_beam(a.n, a.levels, variable_viscosity=True, device=dev)
    return scenes.thin_sheet(a.n, a.levels, thickness_cells=32, device=dev)


if a.source == "oracle":
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
    from util import oracle_for_scene
    from oracle import oracle as O
    dev = torch.device("cpu")
    sc = make_scene(dev)
    o = oracle_for_scene(sc)
    o.prepass(); o.hot_path()
    m = o.csr()
    rp = torch.from_numpy(m.row_ptr.astype(np.int64)); col = torch.from_numpy(m.col.astype(np.int64)); val = torch.from_numpy(m.val)
    tab = torch.from_numpy(o.dof_table(O.I_VELOCITY).astype(np.int64))
    nx, ny, nz = sc.res
    n, nnz = m.n, len(m.col)
else:
    from adaptiveviscositysolver_amd import DevicePrepass, ViscositySolve, capi
    dev = torch.device("cuda:0")
    sc = make_scene(dev)
    pp = DevicePrepass(sc.res, sc.dx, sc.levels)
    pi = pp.run(sc.liquid, sc.solid)
    s = ViscositySolve(sc.res, sc.dx, sc.dt, pi.levels)
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
    rp = rp.long(); col = col.long(); tab = tab.long()

lv = tab[:, 0] & 0xff
ax = (tab[:, 0] >> 8) & 0xff
I = [tab[:, 1 + k] for k in range(3)]
P = [(I[k] << lv) for k in range(3)]
P = [P[0].clamp(max=nx - 1), P[1].clamp(max=ny - 1), P[2].clamp(max=nz - 1)]
nbx, nby = (nx + 7) >> 3, (ny + 7) >> 3
brick = ((P[2] >> 3) * nby + (P[1] >> 3)) * nbx + (P[0] >> 3)
key = (brick << 9) | ((P[2] & 7) << 6) | ((P[1] & 7) << 3) | (P[0] & 7)
perm = torch.sort(key, stable=True).indices
inv = torch.empty_like(perm); inv[perm] = torch.arange(n, device=dev)

# everything below in NEW (brick-major) numbering
lens = (rp[1:] - rp[:-1])[perm]
brick_n = brick[perm]; lv_n = lv[perm]; ax_n = ax[perm]
In = [I[k][perm] for k in range(3)]
rows_old = torch.repeat_interleave(torch.arange(n, device=dev), rp[1:] - rp[:-1])
# order non-zeros by NEW row (stable): entry order inside a row is untouched
row_new_unsorted = inv[rows_old]
order = torch.sort(row_new_unsorted, stable=True).indices
row_new = row_new_unsorted[order]
col_new = inv[col[order]]
uv, code = torch.unique(val, return_inverse=True)
code = code[order]
del rows_old, row_new_unsorted, order, col, val
rp_new = torch.zeros(n + 1, dtype=torch.int64, device=dev); rp_new[1:] = torch.cumsum(lens, 0)
j_in_row = torch.arange(nnz, device=dev) - rp_new[:-1][row_new]

# tile = brick
tile_of_row = brick_n
bx = tile_of_row % nbx; by = (tile_of_row // nbx) % nby; bz = tile_of_row // (nbx * nby)
# multi-level lattices: level l faces at brick-relative level-l cell (-1 .. 8>>l) -> slot LOFF[l] + ((rz*S+ry)*S+rx)*3 + axis
LOFF = [0, 3000, 3648, 3840, 3921]
def lattice_slot(l, axis, i, j, k, obx, oby, obz):
    w = torch.tensor([8, 4, 2, 1, 0], device=dev)[l.clamp(max=4)]
    S = w + 2
    rx, ry, rz = i - w * obx + 1, j - w * oby + 1, k - w * obz + 1
    ok = (l < 4) & (rx >= 0) & (rx < S) & (ry >= 0) & (ry < S) & (rz >= 0) & (rz < S)
    off = torch.tensor(LOFF, device=dev)[l.clamp(max=4)]
    return torch.where(ok, off + ((rz * S + ry) * S + rx) * 3 + axis, torch.full_like(rx, -1))
def base_slot(lr, i, j, k, obx, oby, obz, lc):
    # base of a row (level lr, level-lr cell i,j,k) in the lattice of level lc
    def conv(v, ob):
        wl = torch.tensor([8, 4, 2, 1, 0], device=dev)[lr.clamp(max=4)]
        loc = v - wl * ob                      # local level-lr cell, -1 .. w
        up = (lc - lr).clamp(min=0); dn = (lr - lc).clamp(min=0)
        return ((loc >> up) << dn) + 1
    wc = torch.tensor([8, 4, 2, 1, 0], device=dev)[lc.clamp(max=4)]
    S = wc + 2
    off = torch.tensor(LOFF, device=dev)[lc.clamp(max=4)]
    return off + ((conv(k, obz) * S + conv(j, oby)) * S + conv(i, obx)) * 3
own = lattice_slot(lv_n, ax_n, In[0], In[1], In[2], bx, by, bz)
t_e = tile_of_row[row_new]
lc = lv_n[col_new]
cslot = lattice_slot(lc, ax_n[col_new], In[0][col_new], In[1][col_new], In[2][col_new], bx[row_new], by[row_new], bz[row_new])
extra = cslot < 0
pair = t_e[extra] * n + col_new[extra]
up, upinv = torch.unique(pair, return_inverse=True)
up_tile = up // n
first_of_tile = torch.searchsorted(up_tile, up_tile)
ex_rank = torch.arange(len(up), device=dev) - first_of_tile
cslot[extra] = 3921 + ex_rank[upinv]
extra_per_tile = torch.bincount(up_tile, minlength=int(brick.max()) + 1)
base = base_slot(lv_n[row_new], In[0][row_new], In[1][row_new], In[2][row_new], bx[row_new], by[row_new], bz[row_new], lc)
base = torch.where(extra, torch.zeros_like(base), base)
delta = (cslot - base) * 8 + torch.where(extra, torch.full_like(lc, 4), lc)
M1, M2, M3 = -7046029254386353131, -4417276706812531889, 1609587929392839161
h = ((delta * M1) ^ (code * M2)) * M3
h = (h ^ (h >> 29)) * (2 * j_in_row + 1) * M1
rowh = torch.zeros(n, dtype=torch.int64, device=dev).index_add_(0, row_new, h)
rowh = rowh * 31 + lens
row_has_extra = torch.zeros(n, dtype=torch.bool, device=dev)
row_has_extra[row_new[extra]] = True

ntiles = int(brick.max()) + 1
rows_per_tile = torch.bincount(tile_of_row, minlength=ntiles)
nnz_per_tile = torch.bincount(tile_of_row, weights=lens.double(), minlength=ntiles).long()
live = rows_per_tile > 0
# patterns per tile
th = torch.stack([tile_of_row, rowh], 1)
upat, pinv, pcnt = torch.unique(th, dim=0, return_inverse=True, return_counts=True)
pat_tile = upat[:, 0]
# pattern length = length of any member row
pat_len = torch.zeros(len(upat), dtype=torch.int64, device=dev); pat_len[pinv] = lens
pats_per_tile = torch.bincount(pat_tile, minlength=ntiles)
shared = pcnt[pinv] >= 2
pat_words_per_tile = torch.bincount(pat_tile, weights=pat_len.double(), minlength=ntiles).long()
# global patterns (regular rows only: no extra slot)
gpat = torch.unique(rowh[~row_has_extra])

# fill runs: distinct (tile, slot, column) triples sorted by (tile, slot); a run breaks when slot or column is not previous + 1
trip = torch.unique(torch.stack([t_e, cslot, col_new], 1), dim=0)
brk = torch.ones(len(trip), dtype=torch.bool, device=dev)
brk[1:] = (trip[1:, 0] != trip[:-1, 0]) | (trip[1:, 1] != trip[:-1, 1] + 1) | (trip[1:, 2] != trip[:-1, 2] + 1)
runs_per_tile = torch.bincount(trip[brk, 0], minlength=ntiles)
slots_per_tile = torch.bincount(trip[:, 0], minlength=ntiles)

def q(t, qs=(0.1, 0.5, 0.9, 1.0)):
    t = t.double()
    return [float(torch.quantile(t, x)) if len(t) < 10_000_000 else float(np.quantile(t.cpu().numpy(), x)) for x in qs]

big = rows_per_tile >= 64
out = {
    "scene": a.scene, "n": a.n, "rows": n, "nnz": nnz, "values": int(len(uv)), "tiles": int(live.sum()),
    "tiles_ge512_rows": int(big.sum()), "rows_in_tiles_ge512": int(rows_per_tile[big].sum()),
    "rows_per_tile_p10_50_90_max": q(rows_per_tile[live]),
    "level0_rows": int((lv_n == 0).sum()),
    "rows_without_extra": int((~row_has_extra).sum()),
    "rows_in_shared_patterns": int(shared.sum()),
    "nnz_in_shared_patterns": int(lens[shared].sum()),
    "global_patterns_of_rows_without_extra": int(len(gpat)),
    "patterns_per_tile_p10_50_90_max(ge512)": q(pats_per_tile[big]),
    "pattern_words_per_tile_p10_50_90_max(ge512)": q(pat_words_per_tile[big]),
    "pattern_words_total": int(pat_words_per_tile.sum()),
    "extra_slots_per_tile_p10_50_90_max(ge512)": q(extra_per_tile[big]),
    "slots_per_tile_p10_50_90_max(ge512)": q(slots_per_tile[big]),
    "runs_per_tile_p10_50_90_max(ge512)": q(runs_per_tile[big]),
    "runs_total": int(runs_per_tile.sum()),
    "bytes": {
        "today_4B_per_nnz": 4 * nnz,
        "pattern_id_2B_per_row": 2 * n,
        "pattern_words_4B": 4 * int(pat_words_per_tile.sum()),
        "fill_runs_8B": 8 * int(runs_per_tile.sum()),
    },
}
reg = (~row_has_extra) & big[tile_of_row] & (lens <= 32)
gp, gcnt = torch.unique(rowh[reg], return_counts=True)
gl = torch.zeros(len(gp), dtype=torch.int64, device=dev); gl[torch.searchsorted(gp, rowh[reg])] = lens[reg]
out["regular"] = {"rows": int(reg.sum()), "nnz": int(lens[reg].sum()), "global_patterns": int(len(gp)), "global_pattern_words": int(gl.sum()),
                  "rows_in_tiles_ge64": int(big[tile_of_row].sum()), "max_len_lattice_rows": int(lens[~row_has_extra].max()),
                  "level_hist_rows": torch.bincount(lv_n, minlength=5).tolist(),
                  "streamed_nnz": int(nnz - lens[reg].sum())}
tp = torch.unique(torch.stack([tile_of_row[reg], rowh[reg]], 1), dim=0)
ppt = torch.bincount(tp[:, 0], minlength=ntiles)
out["regular"]["patterns_per_tile_p10_50_90_max"] = q(ppt[big])
out["bytes"]["v1_estimate"] = 4 * n + 4 * out["regular"]["streamed_nnz"] + 8 * int(runs_per_tile.sum()) + 2 * int(ppt.sum())
out["bytes"]["brick_total"] = out["bytes"]["pattern_id_2B_per_row"] + out["bytes"]["pattern_words_4B"] + out["bytes"]["fill_runs_8B"]
print(json.dumps(out, indent=1))
if a.out:
    with open(a.out, "w") as f: json.dump(out, f, indent=1)
