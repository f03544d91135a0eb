"""TEST INFRASTRUCTURE: a second, independent restatement of the pre-pass (everything solveGasSubclass does
BEFORE the hot path, cpp:233-416) as torch tensor ops.  It is not part of the product: the product's pre-pass is
avs_prepass.hip behind avs_prepass_* (include/avs.h).  tests/test_prepass.py requires it to agree bit for bit with
the C oracle's pre-pass on the CPU -- two implementations written differently that must produce the same integers.
It restates:
  * integration weights    cpp:712-766  (HDK computeSDFWeightsSampled: unpinned, defined in
                           oracle/avs_oracle.c `weights_for_lattice`; same definition here)
  * refinement mask        cpp:815-867
  * octree label pyramid   HDK_OctreeGrid.cpp:4-243, 310-920
  * classification         cpp:1087-1443
  * serial numbering       cpp:1445-1715 (HDK 16^3 tile order, x fastest)
All arrays are (nz, ny, nx)-shaped, x fastest.  fp32 arithmetic is done one rounding per
operation in the order of the oracle so results are bit-identical.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import torch

INACTIVE, ACTIVE, UP, DOWN = 0, 1, 2, 3
FLUID, UNASSIGNED, SOLIDBOUNDARY, OUTSIDE = 0, -1, -2, -3
TILE = 16

# dims of a (nz, ny, nx) tensor for axis 0/1/2 = x/y/z
DIM = (2, 1, 0)


@dataclass
class Pyramid:
    """Inputs of the hot path (what cpp:233-416 hands to cpp:418)."""
    res: tuple
    dx: float
    dt: float
    levels: int
    labels: list                      # [level] int8 (nz,ny,nx)>>level
    vidx: list                        # [level][axis] int32 face lattice
    eidx: list                        # [level][axis] int32 edge lattice
    cidx: list                        # [level] int32 centre lattice
    n_velocity: int
    n_edge: int
    n_center: int
    center_weights: torch.Tensor = None
    edge_weights: list = field(default_factory=list)
    face_weights: list = field(default_factory=list)
    mask: torch.Tensor = None


# --------------------------------------------------------------------------------------------
# integration weights
# --------------------------------------------------------------------------------------------
def _subsample_consts(n, target_centered, s):
    d = ((0.5 if target_centered else 0.0) - 0.5) + ((s + 0.5) / n - 0.5)
    fl = math.floor(d)
    return int(fl), d - fl


def _lerp_axis(src, dim, n_target, di, fr):
    """lerp along `dim` onto a target lattice of n_target samples: a*(1-t) + b*t in fp32."""
    n_src = src.shape[dim]
    dev = src.device
    base = torch.arange(n_target, device=dev) + di
    i0 = base.clamp(0, n_src - 1)
    i1 = (base + 1).clamp(0, n_src - 1)
    t = torch.tensor(fr, dtype=torch.float32, device=dev)
    s = torch.tensor(1.0, dtype=torch.float32, device=dev) - t
    a = src.index_select(dim, i0)
    b = src.index_select(dim, i1)
    pa = a * s
    pb = b * t
    return pa + pb


def sdf_weights(sdf, lattice_centered, n_super=3):
    """Fraction of n^3 sub-samples with interpolated SDF < 0 around every sample of a lattice.

    lattice_centered = (cx, cy, cz): True where the target lattice is cell-centred along that axis.
    """
    nz, ny, nx = sdf.shape
    src_n = (nx, ny, nz)
    tgt = [src_n[a] + (0 if lattice_centered[a] else 1) for a in range(3)]
    n = n_super
    count = torch.zeros((tgt[2], tgt[1], tgt[0]), dtype=torch.int32, device=sdf.device)
    for sx in range(n):
        dix, frx = _subsample_consts(n, lattice_centered[0], sx)
        ax = _lerp_axis(sdf, 2, tgt[0], dix, frx)
        for sy in range(n):
            diy, fry = _subsample_consts(n, lattice_centered[1], sy)
            ay = _lerp_axis(ax, 1, tgt[1], diy, fry)
            for sz in range(n):
                diz, frz = _subsample_consts(n, lattice_centered[2], sz)
                az = _lerp_axis(ay, 0, tgt[2], diz, frz)
                count += (az < 0).to(torch.int32)
    return count.to(torch.float32) / torch.tensor(float(n * n * n), dtype=torch.float32, device=sdf.device)


def build_weights(liquid, n_super=3):
    cw = sdf_weights(liquid, (True, True, True), n_super)
    ew = [sdf_weights(liquid, tuple(a == ax for a in range(3)), n_super) for ax in range(3)]
    fw = [sdf_weights(liquid, tuple(a != ax for a in range(3)), n_super) for ax in range(3)]
    return cw, ew, fw


# --------------------------------------------------------------------------------------------
# mask + octree
# --------------------------------------------------------------------------------------------
def build_mask(liquid, solid, dx, extrapolation_scale=0.5):
    sdf = liquid.to(torch.float64)
    extrap = dx * extrapolation_scale
    inner = dx * max(2.0, 0.0)
    outer = 3.0 * dx
    if solid is None:
        sol = torch.full_like(sdf, -1.0)
    else:
        sol = solid.to(torch.float64)
    m = torch.ones_like(liquid, dtype=torch.int8)
    band_out = (sdf > 0) & (sdf < outer)
    neg = sdf <= 0
    near = neg & (sdf > -inner)
    deep = neg & ~near
    deep_solid = deep & (sol > (-inner - extrap))
    m[band_out | near | deep_solid] = 0
    m[deep & ~deep_solid] = -1
    return m


def _blocks(t):
    nz, ny, nx = t.shape
    return t.view(nz // 2, 2, ny // 2, 2, nx // 2, 2)


def _any_blocks(b):
    return _blocks(b).any(dim=5).any(dim=3).any(dim=1)


def _expand(p):
    return p.repeat_interleave(2, 0).repeat_interleave(2, 1).repeat_interleave(2, 2)


def build_octree(mask, desired_levels):
    nz, ny, nx = mask.shape
    L = desired_levels
    for n in (nx, ny, nz):
        L = min(L, int(math.log2(n)))
    L = max(L, 1)
    lab0 = torch.full_like(mask, INACTIVE)
    lab0[mask == 0] = ACTIVE
    lab0[mask < 0] = UP
    labels = [lab0]
    for l in range(L - 1):
        lab = labels[l]
        sz = [s // 2 for s in lab.shape]
        par = torch.full(sz, INACTIVE, dtype=torch.int8, device=lab.device)
        # pass 1
        any_act = _any_blocks(lab == ACTIVE)
        lab = torch.where((lab == UP) & _expand(any_act), torch.full_like(lab, ACTIVE), lab)
        labels[l] = lab
        par[any_act] = DOWN
        # pass 2: DOWN list, then ACTIVE list
        par[_any_blocks(lab == DOWN)] = DOWN
        act = lab == ACTIVE
        nb = torch.zeros_like(act)
        for d in range(3):
            lo = [slice(None)] * 3
            hi = [slice(None)] * 3
            lo[d] = slice(0, -1)
            hi[d] = slice(1, None)
            nb[tuple(lo)] |= act[tuple(hi)]
            nb[tuple(hi)] |= act[tuple(lo)]
        par[_any_blocks((lab == UP) & nb)] = ACTIVE
        # pass 3
        up_par = _any_blocks(lab == UP) & (par == INACTIVE)
        par[up_par] = UP
        labels.append(par)
    top = labels[L - 1]
    top[top == UP] = ACTIVE
    capped = 0
    while capped < L and bool((labels[capped] == ACTIVE).any()):
        capped += 1
    return labels[:capped]


# --------------------------------------------------------------------------------------------
# classification + numbering
# --------------------------------------------------------------------------------------------
def _tile_expand(hit):
    """per-voxel 'my 16^3 tile contains a hit' mask."""
    shp = hit.shape
    pad = [(-s) % TILE for s in shp]
    h = torch.nn.functional.pad(hit.to(torch.uint8), (0, pad[2], 0, pad[1], 0, pad[0])).to(torch.bool)
    tz, ty, tx = h.shape[0] // TILE, h.shape[1] // TILE, h.shape[2] // TILE
    t = h.view(tz, TILE, ty, TILE, tx, TILE).any(dim=5).any(dim=3).any(dim=1)
    e = t.repeat_interleave(TILE, 0).repeat_interleave(TILE, 1).repeat_interleave(TILE, 2)
    return e[:shp[0], :shp[1], :shp[2]]


def _number(grid, start):
    """Serial sweep in HDK tile order: FLUID -> start, start+1, ... (cpp:1566-1593)."""
    shp = grid.shape
    flag = grid == FLUID
    pad = [(-s) % TILE for s in shp]
    f = torch.nn.functional.pad(flag.to(torch.uint8), (0, pad[2], 0, pad[1], 0, pad[0])).to(torch.bool)
    tz, ty, tx = f.shape[0] // TILE, f.shape[1] // TILE, f.shape[2] // TILE
    ft = f.view(tz, TILE, ty, TILE, tx, TILE).permute(0, 2, 4, 1, 3, 5).reshape(-1)
    ids = torch.cumsum(ft.to(torch.int64), 0) - 1 + start
    total = int(ids[-1].item()) + 1 - start if ft.numel() else 0
    ids = torch.where(ft, ids, torch.full_like(ids, -1))
    back = ids.view(tz, ty, tx, TILE, TILE, TILE).permute(0, 3, 1, 4, 2, 5).reshape(f.shape)
    back = back[:shp[0], :shp[1], :shp[2]]
    out = torch.where(flag, back.to(torch.int32), grid)
    return out, start + total


def _sl(dim, s):
    idx = [slice(None)] * 3
    idx[dim] = s
    return tuple(idx)


def _lerp_half(a, b):
    h = torch.tensor(0.5, dtype=torch.float32, device=a.device)
    s = torch.tensor(1.0, dtype=torch.float32, device=a.device) - h
    return a * s + b * h


def classify_velocity(labels, liquid, solid, cw, ew, dx, extrapolation_scale=0.5):
    L = len(labels)
    extrap = dx * extrapolation_scale
    dev = liquid.device
    out = []
    for l in range(L):
        lab = labels[l]
        per_axis = []
        for axis in range(3):
            d = DIM[axis]
            shp = list(lab.shape)
            shp[d] += 1
            g = torch.full(shp, UNASSIGNED, dtype=torch.int32, device=dev)
            hit = (liquid.to(torch.float64) < 2.0 * dx) if l == 0 else (lab == ACTIVE)
            fh = torch.zeros(shp, dtype=torch.bool, device=dev)
            fh[_sl(d, slice(0, -1))] |= hit
            fh[_sl(d, slice(1, None))] |= hit
            occ = _tile_expand(fh)
            interior = _sl(d, slice(1, -1))
            bl = lab[_sl(d, slice(0, -1))]
            fl = lab[_sl(d, slice(1, None))]
            gi = g[interior]
            if l == 0:
                g[_sl(d, slice(0, 1))] = OUTSIDE
                g[_sl(d, slice(-1, None))] = OUTSIDE
                both = (bl == ACTIVE) & (fl == ACTIVE)
                active = (cw[_sl(d, slice(0, -1))] > 0) | (cw[_sl(d, slice(1, None))] > 0)
                for ea in range(3):
                    if ea == axis:
                        continue
                    oa = 3 - axis - ea          # offset axis of HDKfaceToEdge
                    do = DIM[oa]
                    e = ew[ea][interior]        # edge lattice ea has +1 on `axis` and `oa`; interior faces <-> 1..n-1 on axis
                    active = active | (e[_sl(do, slice(0, -1))] > 0) | (e[_sl(do, slice(1, None))] > 0)
                if solid is None:
                    is_solid = torch.zeros_like(both) if (-1.0 <= -extrap) else torch.ones_like(both)
                else:
                    sv = _lerp_half(solid[_sl(d, slice(0, -1))], solid[_sl(d, slice(1, None))])
                    is_solid = sv.to(torch.float64) > -extrap
                val = torch.full_like(gi, UNASSIGNED)
                val = torch.where(both & active & is_solid, torch.full_like(val, SOLIDBOUNDARY), val)
                val = torch.where(both & active & ~is_solid, torch.full_like(val, FLUID), val)
                val = torch.where(both & ~active, torch.full_like(val, OUTSIDE), val)
                inact = ~both & ((bl == INACTIVE) | (fl == INACTIVE))
                val = torch.where(inact, torch.full_like(val, OUTSIDE), val)
                trans = ~both & ~inact & (((bl == UP) & (fl == ACTIVE)) | ((bl == ACTIVE) & (fl == UP)))
                val = torch.where(trans, torch.full_like(val, FLUID), val)
            else:
                fluid = ((bl == ACTIVE) & (fl == ACTIVE)) | ((bl == UP) & (fl == ACTIVE)) | ((bl == ACTIVE) & (fl == UP))
                val = torch.where(fluid, torch.full_like(gi, FLUID), torch.full_like(gi, UNASSIGNED))
            g[interior] = val
            g = torch.where(occ, g, torch.full_like(g, UNASSIGNED))
            per_axis.append(g)
        out.append(per_axis)
    return out


def classify_edges(labels, ew):
    L = len(labels)
    out = []
    for l in range(L):
        lab = labels[l]
        dev = lab.device
        per_axis = []
        for axis in range(3):
            a1, a2 = (axis + 1) % 3, (axis + 2) % 3
            d1, d2 = DIM[a1], DIM[a2]
            shp = list(lab.shape)
            shp[d1] += 1
            shp[d2] += 1
            # occupied tiles: edges of ACTIVE cells (cpp:1003-1057)
            act = lab == ACTIVE
            eh = torch.zeros(shp, dtype=torch.bool, device=dev)
            for o1 in (0, 1):
                for o2 in (0, 1):
                    idx = [slice(None)] * 3
                    idx[d1] = slice(o1, shp[d1] - 1 + o1)
                    idx[d2] = slice(o2, shp[d2] - 1 + o2)
                    eh[tuple(idx)] |= act
            occ = _tile_expand(eh)
            # pad labels by one sentinel layer on the two perpendicular axes
            padspec = [0, 0, 0, 0, 0, 0]  # (x_lo, x_hi, y_lo, y_hi, z_lo, z_hi) for F.pad order (last dim first)
            for a in (a1, a2):
                padspec[2 * a] = 1
                padspec[2 * a + 1] = 1
            lp = torch.nn.functional.pad(lab, tuple(padspec), value=9)
            val = torch.full(shp, UNASSIGNED, dtype=torch.int32, device=dev)
            active = torch.zeros(shp, dtype=torch.bool, device=dev)
            stopped = torch.zeros(shp, dtype=torch.bool, device=dev)
            for ci in range(4):  # HDKedgeToCell: bit clear -> -1 on that axis
                o1 = 0 if (ci & 1) else -1
                o2 = 0 if (ci & 2) else -1
                idx = [slice(None)] * 3
                # edge e (0..n) -> padded cell index e + o + 1
                idx[d1] = slice(o1 + 1, o1 + 1 + shp[d1])
                idx[d2] = slice(o2 + 1, o2 + 1 + shp[d2])
                lc = lp[tuple(idx)]
                oob = (lc == 9) & ~stopped
                val = torch.where(oob, torch.full_like(val, OUTSIDE), val)
                stopped = stopped | oob
                down = (lc == DOWN) & ~stopped
                active = active & ~down
                stopped = stopped | down
                active = active | ((lc == ACTIVE) & ~stopped)
            if l == 0:
                fin = torch.where(ew[axis] > 0, torch.full_like(val, FLUID), torch.full_like(val, OUTSIDE))
            else:
                fin = torch.full_like(val, FLUID)
            val = torch.where(active, fin, val)
            val = torch.where(occ, val, torch.full_like(val, UNASSIGNED))
            per_axis.append(val)
        out.append(per_axis)
    return out


def classify_centers(labels, cw):
    out = []
    for l, lab in enumerate(labels):
        g = torch.full(lab.shape, UNASSIGNED, dtype=torch.int32, device=lab.device)
        sel = (lab == ACTIVE) if l != 0 else ((lab == ACTIVE) & (cw > 0))
        g[sel] = FLUID
        out.append(g)
    return out


def build_pyramid(scene, n_super=3, extrapolation_scale=0.5) -> Pyramid:
    """Run the whole pre-pass for a scenes.Scene; tensors stay on scene.liquid.device."""
    liquid = scene.liquid
    cw, ew, fw = build_weights(liquid, n_super)
    mask = build_mask(liquid, scene.solid, scene.dx, extrapolation_scale)
    labels = build_octree(mask, scene.levels)
    vidx = classify_velocity(labels, liquid, scene.solid, cw, ew, scene.dx, extrapolation_scale)
    eidx = classify_edges(labels, ew)
    cidx = classify_centers(labels, cw)
    nv = ne = nc = 0
    for l in range(len(labels)):
        for a in range(3):
            vidx[l][a], nv = _number(vidx[l][a], nv)
    for l in range(len(labels)):
        for a in range(3):
            eidx[l][a], ne = _number(eidx[l][a], ne)
    for l in range(len(labels)):
        cidx[l], nc = _number(cidx[l], nc)
    return Pyramid(res=scene.res, dx=scene.dx, dt=scene.dt, levels=len(labels), labels=labels, vidx=vidx,
                   eidx=eidx, cidx=cidx, n_velocity=nv, n_edge=ne, n_center=nc, center_weights=cw,
                   edge_weights=ew, face_weights=fw, mask=mask)
