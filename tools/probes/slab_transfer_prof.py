"""One rank of an 8-way slab-local frame alone, its transfer five times (for rocprofv3 --kernel-trace --stats): N LEVELS SCENE RANK"""
import ctypes as C
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from adaptiveviscositysolver_amd import DevicePrepass, ViscositySolve, capi, scenes  # noqa: E402
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
levels = int(sys.argv[2]) if len(sys.argv) > 2 else 4
scene = sys.argv[3] if len(sys.argv) > 3 else "beam"
rank = int(sys.argv[4]) if len(sys.argv) > 4 else 3
world, axis = 8, 0
dev = torch.device("cuda:0")
sc = {"beam": lambda: scenes.fat_beam(n, levels, device=dev), "sheet": lambda: scenes.thin_sheet(n, levels, thickness_cells=32, device=dev)}[scene]()
cuts = np.asarray([n * r // world for r in range(world + 1)], np.int32)
hip = C.CDLL("libamdhip64.so")
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
hip.hipStreamSynchronize.argtypes = [C.c_void_p]
parts = []


def record(ptr, count, stream):
    hip.hipStreamSynchronize(C.c_void_p(stream))
    mine = np.empty(count, np.int32)
    hip.hipMemcpy(mine.ctypes.data, C.c_void_p(ptr), count * 4, 2)
    parts.append(mine.astype(np.int64))


for r in range(world):
    p = DevicePrepass(sc.res, sc.dx, sc.levels)
    p.set_slab(axis, cuts, r, record)
    p.run(sc.liquid, sc.solid)
    p.close()
recorded = np.sum(parts, axis=0).astype(np.int32)


def replay(ptr, count, stream):
    hip.hipStreamSynchronize(C.c_void_p(stream))
    hip.hipMemcpy(C.c_void_p(ptr), recorded.ctypes.data, count * 4, 1)


p = DevicePrepass(sc.res, sc.dx, sc.levels)
p.set_slab(axis, cuts, rank, replay)
info = p.run(sc.liquid, sc.solid)
s = ViscositySolve(sc.res, sc.dx, sc.dt, info.levels, device=0)
s.dist_init_hosted(rank, world)
p.apply(s)
s.set_scene_fields(sc)
capi.check(s.lib.avs_dist_assemble(s.h, axis, None))
s.set_solution(torch.linspace(0., 1., int(info.n_velocity), dtype=torch.float64, device=dev))
vel = [v.clone() for v in sc.velocity]
import time
for i in range(6):
    torch.cuda.synchronize(); t = time.perf_counter()
    s.transfer_to_regular_grid_in_place(vel)
    torch.cuda.synchronize(); print("transfer", i, round((time.perf_counter() - t) * 1e3, 3), flush=True)
