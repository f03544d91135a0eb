"""Velocity DOFs per octree level of the headline scene (how much of the restriction / row work sits on coarse faces)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from adaptiveviscositysolver_amd import DevicePrepass, scenes
n0 = int(sys.argv[1]) if len(sys.argv) > 1 else 512
sc = scenes.fat_beam(n0, 4, variable_viscosity=False, device=torch.device("cuda", 0))
pp = DevicePrepass(sc.res, sc.dx, sc.levels, device=0)
info = pp.run(sc.liquid, sc.solid)
tot = 0
for lv in range(info.levels):
    c = sum(int((pp.index(0, lv, a) >= 0).sum()) for a in range(3))
    tot += c
    print("level", lv, "velocity dofs", c, "restriction leaves", c * 12 ** lv)
print("total", tot, "of", info.n_velocity)
