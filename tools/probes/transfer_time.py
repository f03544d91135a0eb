"""Post-solve transfer (avs_transfer_to_regular_grid), device destination: wall time of the first and of later calls."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from adaptiveviscositysolver_amd import DevicePrepass, ViscositySolve, capi, scenes
dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "beam"
sc = scenes.fat_beam(512, 4, device=dev) if which == "beam" else scenes.thin_sheet(1024, 5, thickness_cells=32, device=dev)
pp = DevicePrepass(sc.res, sc.dx, sc.levels); pi = pp.run(sc.liquid, sc.solid)
s = ViscositySolve(sc.res, sc.dx, sc.dt, pi.levels); pp.apply(s); s.set_scene_fields(sc); s.assemble()
s.solve(1e-3, 2500)
outs = [torch.empty_like(v) for v in sc.velocity]
for i in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    capi.check(s.lib.avs_transfer_to_regular_grid(s.h, outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(), capi.MEM_DEVICE))
    torch.cuda.synchronize(); print(which, "transfer call", i, "%.2f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
