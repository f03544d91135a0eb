"""How many column indices of a 256-row SpMV tile fall inside a window of x entries around the tile's rows
(brick-major numbering, read through the 1-rank distributed plan)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from adaptiveviscositysolver_amd import DevicePrepass, ViscositySolve, capi, scenes

n, levels = int(sys.argv[1]) if len(sys.argv) > 1 else 256, 4
dev = torch.device("cuda:0")
sc = scenes.fat_beam(n, levels, device=dev)
pp = DevicePrepass(sc.res, sc.dx, sc.levels)
pi = pp.run(sc.liquid, sc.solid)
s = ViscositySolve(sc.res, sc.dx, sc.dt, pi.levels, device=0)
pp.apply(s); pp.close()
s.set_scene_fields(sc)
s.assemble()
buf = (C.c_uint8 * capi.UNIQUE_ID_BYTES)()
capi.check(s.lib.avs_dist_get_unique_id(buf))
capi.check(s.lib.avs_dist_init(s.h, buf, 0, 1))
sz = s.dist_partition()
rp = np.empty(sz.n_own + 1, np.int32); col = np.empty(sz.nnz_local, np.int32)
capi.check(s.lib.avs_dist_get_plan_arrays(s.h, None, rp.ctypes.data, col.ctypes.data, None, None, None, None, None, None))
rows = np.repeat(np.arange(sz.n_own, dtype=np.int64), np.diff(rp))
for T in (256, 512):
    tile0 = (rows // T) * T
    for W in (T, 2 * T, 4 * T, 8 * T):
        lo = tile0 - (W - T) // 2
        inside = (col >= lo) & (col < lo + W)
        print(f"tile {T} window {W}: {inside.mean()*100:.1f}% of the columns inside")
d = np.abs(col.astype(np.int64) - rows)
for q in (1, 8, 64, 256, 512, 1536, 4096, 100000):
    print(f"|col-row| <= {q}: {(d <= q).mean()*100:.1f}%")
