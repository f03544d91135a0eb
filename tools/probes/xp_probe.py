"""Round 6 cost probes of the brick kernel on the 512^3 beam, one process, one box: phase switches, the fused direction update (AVS_BRICK_XP:
extra residual / code / solution streams in the fill -- wrong y on purpose), s_setprio around the load phase.  Warm = back-to-back launches,
cold = 600 MB overwritten between launches (as inside the PCG loop)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from adaptiveviscositysolver_amd import DevicePrepass, ViscositySolve, scenes
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
sc = scenes.fat_beam(n, 4, device=dev)
pp = DevicePrepass(sc.res, sc.dx, sc.levels); pi = pp.run(sc.liquid, sc.solid)
s = ViscositySolve(sc.res, sc.dx, sc.dt, pi.levels, probe=True); pp.apply(s); s.set_scene_fields(sc); pp.close(); s.assemble()
print(s.matrix_format().brick_tiles, "tiles")
def run(tag, dbg=0, xp=0, cold=False, reps=200):
    os.environ["AVS_BRICK_DEBUG"] = str(dbg)
    if xp: os.environ["AVS_BRICK_XP"] = str(xp)
    else: os.environ.pop("AVS_BRICK_XP", None)
    if cold: os.environ["AVS_BENCH_THRASH_MB"] = "600"
    else: os.environ.pop("AVS_BENCH_THRASH_MB", None)
    ts = [s.bench_spmv(100, reps if not cold else 60) * 1e3 for _ in range(3)]
    print(f"{tag:40s} {'cold' if cold else 'warm'}  us: " + " ".join(f"{t:7.1f}" for t in ts), flush=True)
for cold in (False, True):
    run("baseline", 0, 0, cold)
    run("setprio(2) in the load phase", 512, 0, cold)
    run("XP=1 own rows: r, code, x streams", 0, 1, cold)
    run("XP=2 + halo r, code gathers", 0, 2, cold)
    run("XP=2 + setprio", 512, 2, cold)
    run("baseline again", 0, 0, cold)
