"""Slab-local pre-pass + assembly against the replicated ones, rank by rank, each rank ALONE on the GPU (tools/probes/slab_time.py N LEVELS WORLD [AXIS] [SCENE]).

Virtual ranks share one GPU, so a concurrent run would time the contention.  Instead: (1) one rank of a hosted group runs the replicated
avs_dist_assemble on the whole pre-pass -> the cuts; (2) every rank runs the slab-local pre-pass once with a callback that records its contribution; the recordings are summed; (3) every rank then runs alone, its all-reduce callback replaying the recorded sums (a 1-3 MB host-to-device copy stands in for the
collective), three frames (the last one is the steady state the single-rank numbers are quoted for), then avs_prepass_apply + avs_dist_assemble
as member `r` of a hosted group.  Output: one JSON line per rank + a summary."""
import ctypes as C
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from adaptiveviscositysolver_amd import DevicePrepass, ViscositySolve, capi, scenes  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
levels = int(sys.argv[2]) if len(sys.argv) > 2 else 4
world = int(sys.argv[3]) if len(sys.argv) > 3 else 8
axis = int(sys.argv[4]) if len(sys.argv) > 4 else 0
scene = sys.argv[5] if len(sys.argv) > 5 else "beam"
dev = torch.device("cuda:0")
sc = {"beam": lambda: scenes.fat_beam(n, levels, device=dev), "sheet": lambda: scenes.thin_sheet(n, levels, thickness_cells=32, device=dev),
      "tank": lambda: scenes.tank(n, levels, device=dev)}[scene]()
hip = C.CDLL("libamdhip64.so")
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
hip.hipStreamSynchronize.argtypes = [C.c_void_p]


def phases(info):
    return {"weights": round(info.weights_ms, 3), "octree": round(info.octree_ms, 3), "classify": round(info.classify_ms, 3), "numbering": round(info.number_ms, 3),
            "sum": round(info.weights_ms + info.octree_ms + info.classify_ms + info.number_ms, 3)}


XSTAND = []


def assemble_alone(pp, r, lv):
    s = ViscositySolve(sc.res, sc.dx, sc.dt, lv, device=0)
    s.dist_init_hosted(r, world)
    pp.apply(s)
    s.set_scene_fields(sc)
    capi.check(s.lib.avs_dist_assemble(s.h, axis, None))   # warm-up (code objects, first touch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ai = capi.AssemblyInfo()
    capi.check(s.lib.avs_dist_assemble(s.h, axis, C.byref(ai)))
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e3
    sz = capi.PlanSizes()
    capi.check(s.lib.avs_dist_get_plan_sizes(s.h, C.byref(sz)))
    ax, cuts = s.dist_cuts(world, 0)
    # the post-solve transfer (in place, steady state: the third call) on a stand-in solution -- its time does not depend on the values
    s.set_solution(XSTAND[0] if XSTAND else XSTAND.append(torch.linspace(0., 1., int(ai.n_velocity), dtype=torch.float64, device=dev)) or XSTAND[0])
    vel = [v.clone() for v in sc.velocity]
    tr = 0.
    for _ in range(3):
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        s.transfer_to_regular_grid_in_place(vel)
        torch.cuda.synchronize()
        tr = (time.perf_counter() - t1) * 1e3
    del vel
    out = {"stencils_ms": round(ai.stencil_ms, 3), "rows_ms": round(ai.system_ms, 3), "forms_ms": round(ai.csr_ms, 3), "wall_ms": round(wall, 3),
           "transfer_in_place_ms": round(tr, 3), "n_own": int(sz.n_own), "n_halo": int(sz.n_halo), "nnz_local": int(sz.nnz_local)}
    s.close()
    return out, cuts


# (1) the whole pre-pass, steady state; the replicated assembly of every rank alone
pp = DevicePrepass(sc.res, sc.dx, sc.levels)
for _ in range(3):
    info = pp.run(sc.liquid, sc.solid)
lv = info.levels
full = phases(info)
n_dofs = (info.n_velocity, info.n_edge, info.n_center)
print(json.dumps({"scene": scene, "n": n, "levels": lv, "world": world, "axis": axis, "dofs": n_dofs, "full_prepass_ms": full}), flush=True)
rep = []
cuts = None
for r in range(world):
    o, cuts = assemble_alone(pp, r, lv)
    rep.append(o)
    print(json.dumps({"rank": r, "replicated": o}), flush=True)
pp.close()
torch.cuda.empty_cache()

# (2) the summed exchange buffer: every rank alone, its callback records what it would contribute (and hands its own array back: the rest
# of that run numbers with the rank's counts only -- discarded); the sum of the recordings is what the collective would deliver
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
    torch.cuda.empty_cache()
recorded = np.sum(parts, axis=0).astype(np.int32)


def replay(ptr, count, stream):
    assert count == len(recorded)
    hip.hipStreamSynchronize(C.c_void_p(stream))
    hip.hipMemcpy(C.c_void_p(ptr), recorded.ctypes.data, count * 4, 1)


# (3) every rank alone
loc = []
for r in range(world):
    p = DevicePrepass(sc.res, sc.dx, sc.levels)
    p.set_slab(axis, cuts, r, replay)
    for _ in range(3):
        info = p.run(sc.liquid, sc.solid)
    assert (info.n_velocity, info.n_edge, info.n_center) == n_dofs and info.levels == lv
    lo, hi, nw = p.window()
    o, _ = assemble_alone(p, r, lv)
    rec = {"rank": r, "slab": [int(cuts[r]), int(cuts[r + 1])], "window_level0": [int(lo[0]), int(hi[0])], "window_dofs": nw,
           "window_fraction_of_velocity_dofs": round(nw[0] / max(n_dofs[0], 1), 4), "prepass_ms": phases(info), "assembly": o}
    loc.append(rec)
    print(json.dumps(rec), flush=True)
    assert o["n_own"] == rep[r]["n_own"] and o["nnz_local"] == rep[r]["nnz_local"] and o["n_halo"] == rep[r]["n_halo"]
    p.close()
    torch.cuda.empty_cache()

idx = lambda o: o["stencils_ms"] + o["rows_ms"]
summary = {"exchange_int32": int(len(recorded)), "full_prepass_ms": full["sum"],
           "slab_prepass_ms_max": max(l["prepass_ms"]["sum"] for l in loc), "slab_prepass_ms_mean": round(float(np.mean([l["prepass_ms"]["sum"] for l in loc])), 3),
           "replicated_stencils_plus_rows_ms_max": round(max(idx(o) for o in rep), 3), "slab_stencils_plus_rows_ms_max": round(max(idx(l["assembly"]) for l in loc), 3),
           "replicated_assembly_wall_ms_max": max(o["wall_ms"] for o in rep), "slab_assembly_wall_ms_max": max(l["assembly"]["wall_ms"] for l in loc),
           "whole_transfer_in_place_ms": max(o["transfer_in_place_ms"] for o in rep), "slab_transfer_in_place_ms_max": max(l["assembly"]["transfer_in_place_ms"] for l in loc),
           "window_fraction_max": max(l["window_fraction_of_velocity_dofs"] for l in loc), "ideal_fraction": round(1.0 / world, 4)}
print(json.dumps({"summary": summary}), flush=True)
