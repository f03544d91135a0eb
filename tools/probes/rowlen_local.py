"""Row-length histogram of one rank's local system of the W-way partition (planning aid for the CU-resident loop)."""
import os, sys
os.environ["AVS_DIST_LOOPBACK"] = "1"
os.environ["AVS_CG_RESIDENT"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from adaptiveviscositysolver_amd import DevicePrepass, ViscositySolve, capi, scenes
W, R = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda:0")
sc = scenes.fat_beam(512, 4, device=dev)
pp = DevicePrepass(sc.res, sc.dx, sc.levels); pi = pp.run(sc.liquid, sc.solid)
s = ViscositySolve(sc.res, sc.dx, sc.dt, pi.levels); pp.apply(s); s.set_scene_fields(sc); pp.close()
capi.check(s.lib.avs_dist_init_hosted(s.h, R, W)); s.dist_assemble()
sz = s.plan_sizes
rp = np.empty(sz.n_own + 1, np.int32)
capi.check(s.lib.avs_dist_get_plan_arrays(s.h, None, rp.ctypes.data, None, None, None, None, None, None, None))
L = np.diff(rp)
print("rows", len(L), "nnz", int(rp[-1]), "mean", L.mean())
print("hist", np.bincount(L)[:80].tolist())
for SW, S in ((15, 5), (16, 4), (18, 4), (19, 4), (20, 4), (24, 3), (12, 6), (10, 8), (9, 8), (8, 10), (6, 12), (5, 15)):
    k = -(-L // SW)
    # greedy lanes
    lanes = 0; used = 0
    for v in k.tolist():
        if v > S: lanes += 1 + (1 if used else 0); used = 0; continue
        if used + v > S: lanes += 1; used = 0
        used += v
    lanes += 1 if used else 0
    print(f"slot {SW} x {S} = {SW*S} words: slots/row {k.mean():.3f}, lanes {lanes} ({lanes/262144:.3f} of the chip)")
print("quad slots (5 words), rows per lane capped:")
k = (-(-L // 5)).tolist()
for Q in (14, 15, 16):
    for RM in (5, 6, 7, 8, 15):
        lanes = 0; used = 0; rows = 0
        for v in k:
            if v > Q: lanes += 1 + (1 if used else 0); used = 0; rows = 0; continue
            if used + v > Q or rows >= RM: lanes += 1; used = 0; rows = 0
            used += v; rows += 1
        lanes += 1 if used else 0
        print(f"  {Q} quads, <= {RM} rows: lanes {lanes} ({lanes/262144:.3f}), rows/lane {len(k)/lanes:.2f}")
