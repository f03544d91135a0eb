"""Histogram of the merged row lengths of the headline system (which k_merge_rows path the rows take)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from adaptiveviscositysolver_amd import DevicePrepass, ViscositySolve, scenes
dev = torch.device("cuda", 0)
sc = scenes.fat_beam(512, 4, variable_viscosity=False, device=dev)
pp = DevicePrepass(sc.res, sc.dx, sc.levels, device=0)
pinfo = pp.run(sc.liquid, sc.solid)
s = ViscositySolve(sc.res, sc.dx, sc.dt, pinfo.levels, device=0)
pp.apply(s); s.set_scene_fields(sc); pp.close()
ai = s.assemble()
rp, col, val, rhs = s.csr()
ln = np.diff(rp.astype(np.int64))
print("rows", len(ln), "nnz", int(rp[-1]), "raw", ai.raw_triplets, "mean", ln.mean())
for lo, hi in ((0, 8), (8, 16), (16, 24), (24, 28), (28, 32), (32, 48), (48, 64), (64, 10**9)):
    print("unique in [%d,%d): %d rows" % (lo, hi, int(((ln >= lo) & (ln < hi)).sum())))
w = ln[: len(ln) // 64 * 64].reshape(-1, 64)
print("waves with a row of >= 28 unique entries:", int((w.max(1) >= 28).sum()), "of", len(w))
print("rows >= 28 unique per such wave (mean):", float((w >= 28).sum(1)[w.max(1) >= 28].mean()))
