import sys, os
sys.path.insert(0, "/root/repo")
import torch
from adaptiveviscositysolver_amd import DevicePrepass, ViscositySolve, scenes
dev = torch.device("cuda:0")
sc = scenes.fat_beam(512, 4, device=dev)
pp = DevicePrepass(sc.res, sc.dx, sc.levels); pi = pp.run(sc.liquid, sc.solid)
s = ViscositySolve(sc.res, sc.dx, sc.dt, pi.levels); pp.apply(s); s.set_scene_fields(sc); pp.close(); s.assemble()
for r in range(3): print("default SpMV us:", s.bench_spmv(0, 200) * 1e3)
