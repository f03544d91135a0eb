"""Per-rank iteration time of the W-way partitioned solve, measured on ONE GPU (no multi-GPU box was available).

For world sizes 2, 4, 8 every rank r of the slab partition is run ALONE on cuda:0 as a hosted group member whose peers are
looped back onto itself (AVS_DIST_LOOPBACK=1, avs_dist.hip: direct_connect_loopback): it assembles exactly the rows rank r
would own, and each iteration runs the real update + push (stores into its own block), the interior tiles, the halo-touching
tiles (gathering the never-written, zero halo) and the finalisation with W contributions.  The linear system it solves is NOT
the right one (zero halo); what is representative is the time per iteration of that slab: compute + launch + the intra-GPU
part of the synchronisation, without the xGMI latencies.  Projection: it/s(W) ~ 1 / max_r time_r."""
import argparse
import ctypes as C
import json
import os
import sys

os.environ["AVS_DIST_LOOPBACK"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from adaptiveviscositysolver_amd import DevicePrepass, ViscositySolve, capi, scenes

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=512)
ap.add_argument("--levels", type=int, default=4)
ap.add_argument("--worlds", default="1,2,4,8")
ap.add_argument("--iters", type=int, default=320)
ap.add_argument("--ranks", default="", help="comma-separated subset of ranks to run (profiling one rank under rocprofv3)")
a = ap.parse_args()
only = [int(r) for r in a.ranks.split(",")] if a.ranks else None
if "," in a.worlds:
    # one PROCESS per world size (as a deployment has): dozens of contexts built and freed in one process leave the allocator in a
    # state where the partitioned assembly of the later worlds shows 40-60 ms instead of 10
    import subprocess
    merged = None
    for w in a.worlds.split(","):
        cmd = [sys.executable, os.path.abspath(__file__), "--n", str(a.n), "--levels", str(a.levels), "--worlds", w, "--iters", str(a.iters)]
        if a.ranks:
            cmd += ["--ranks", a.ranks]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
        if r.returncode != 0:
            sys.exit(r.returncode)
        d = json.loads(r.stdout[r.stdout.index("{"):])
        if merged is None:
            merged = d
        else:
            merged["worlds"].update(d["worlds"])
    base = merged["worlds"].get("1", {}).get("projected_iter_per_s")
    for w, v in merged["worlds"].items():
        if base:
            v["projected_speedup_vs_world1_direct"] = v["projected_iter_per_s"] / base
    print(json.dumps(merged, indent=1))
    sys.exit(0)
dev = torch.device("cuda:0")
sc = scenes.fat_beam(a.n, a.levels, device=dev)
pp = DevicePrepass(sc.res, sc.dx, sc.levels)
pi = pp.run(sc.liquid, sc.solid)
out = {"n": a.n, "levels": int(pi.levels), "rows": int(pi.n_velocity), "iterations_timed": a.iters, "worlds": {}}
for world in [int(w) for w in a.worlds.split(",")]:
    ranks = []
    for r in range(world):
        if only is not None and r not in only:
            continue
        s = ViscositySolve(sc.res, sc.dx, sc.dt, pi.levels)
        pp.apply(s)
        s.set_scene_fields(sc)
        capi.check(s.lib.avs_dist_init_hosted(s.h, r, world))
        s.dist_assemble()                             # warm-up pass (first touch), then the timed one
        torch.cuda.synchronize()
        import time
        asm_ms = 1e30                                 # best of three: this process builds and frees dozens of contexts, and an
        for _ in range(3):                            # assembly that lands behind a large free shows 45-60 ms instead of 10
            t_as = time.perf_counter()
            s.dist_assemble()
            torch.cuda.synchronize()
            asm_ms = min(asm_ms, (time.perf_counter() - t_as) * 1e3)
        capi.check(s.lib.avs_dist_import_blobs(s.h, None))
        s.dist_solve(1e-30, 64)                       # warm-up: graph capture, first touch
        info = s.dist_solve(1e-30, a.iters)           # never converges: exactly a.iters iterations
        sz = s.plan_sizes
        ranks.append({"rank": r, "n_own": int(sz.n_own), "n_halo": int(sz.n_halo), "n_send": int(sz.n_send), "n_peers": int(sz.n_peers),
                      "nnz_local": int(sz.nnz_local), "tiles": list(s.overlap_tiles), "us_per_iteration": info.solve_ms * 1e3 / max(info.iterations, 1),
                      "spmv_us": info.spmv_ms * 1e3, "iterations": int(info.iterations), "resident_loop": bool(info.resident),
                      "dist_assemble_wall_ms": asm_ms})
        s.close()
    worst = max(x["us_per_iteration"] for x in ranks)
    # end to end for the headline solve (1271 iterations at tol 1e-3): replicated pre-pass + this rank's distributed assembly + iterations
    pre_ms = pi.weights_ms + pi.octree_ms + pi.classify_ms + pi.number_ms
    e2e = max(pre_ms + x["dist_assemble_wall_ms"] for x in ranks) + 1271 * worst * 1e-3
    out["worlds"][world] = {"ranks": ranks, "max_us_per_iteration": worst, "projected_iter_per_s": 1e6 / worst,
                            "prepass_ms_replicated": pre_ms, "projected_end_to_end_ms_1271_iterations": e2e}
base = out["worlds"].get(1, {}).get("projected_iter_per_s")
for w, v in out["worlds"].items():
    if base:
        v["projected_speedup_vs_world1_direct"] = v["projected_iter_per_s"] / base
print(json.dumps(out, indent=1))
