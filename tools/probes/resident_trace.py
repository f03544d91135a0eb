import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from adaptiveviscositysolver_amd import DevicePrepass, ViscositySolve, capi, scenes
dev = torch.device("cuda:0")
sc = scenes.sphere(32, 3, device=dev)
pp = DevicePrepass(sc.res, sc.dx, sc.levels); pi = pp.run(sc.liquid, sc.solid)
s = ViscositySolve(sc.res, sc.dx, sc.dt, pi.levels); pp.apply(s); s.set_scene_fields(sc); pp.close(); s.assemble()
os.environ["AVS_CG_RESIDENT"] = "1"; os.environ["AVS_CG_RESIDENT_TIMERS"] = "80"; os.environ["AVS_CG_RESIDENT_VERBOSE"] = "2"
info = s.solve(1e-10, 5000); print("resident", info.iterations, info.converged, info.error, info.rhs_norm2)
