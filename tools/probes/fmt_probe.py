import sys
sys.path.insert(0, "/root/repo")
import torch
from adaptiveviscositysolver_amd import DevicePrepass, ViscositySolve, scenes
dev = torch.device("cuda:0")
for n, vv in ((256, True), (512, True)):
    sc = scenes.fat_beam(n, 4, variable_viscosity=vv, device=dev)
    pp = DevicePrepass(sc.res, sc.dx, sc.levels); pi = pp.run(sc.liquid, sc.solid)
    s = ViscositySolve(sc.res, sc.dx, sc.dt, pi.levels); pp.apply(s); s.set_scene_fields(sc); pp.close(); ai = s.assemble()
    f = s.matrix_format()
    print(n, "rows", ai.n_velocity, "nnz", ai.nnz, {k: getattr(f, k) for k in dir(f) if not k.startswith("_")})
    s.close()
