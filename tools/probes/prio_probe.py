"""Round 6: the s_setprio placements of the brick kernel switched off one by one (probe build: AVS_BRICK_DEBUG bit 512 = no priority for the
load phase, 2048 = none for the longest rows' waves) and the load balance of the row walk across a workgroup's waves; one process, one box.
  python tools/probes/prio_probe.py [512] ; SPMV_SCENE=beam|sheet|tank"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from adaptiveviscositysolver_amd import DevicePrepass, ViscositySolve, scenes, capi
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
kind = os.environ.get("SPMV_SCENE", "beam")
sc = (scenes.thin_sheet(n, 4 if n <= 512 else 5, thickness_cells=32, device=dev) if kind == "sheet" else scenes.tank(n, 4, device=dev) if kind == "tank" else scenes.fat_beam(n, 4, device=dev))
pp = DevicePrepass(sc.res, sc.dx, sc.levels); pi = pp.run(sc.liquid, sc.solid)
s = ViscositySolve(sc.res, sc.dx, sc.dt, pi.levels, probe=True); pp.apply(s); s.set_scene_fields(sc); pp.close(); s.assemble()
print(kind, n, s.matrix_format().brick_tiles, "tiles", flush=True)
out = (C.c_double * 6)()
capi.check(s.lib.avs_brick_wave_stats(s.h, out))
def run(tag, dbg=0, cold=False, reps=200):
    os.environ["AVS_BRICK_DEBUG"] = str(dbg)
    if cold: os.environ["AVS_BENCH_THRASH_MB"] = "600"
    else: os.environ.pop("AVS_BENCH_THRASH_MB", None)
    ts = [s.bench_spmv(100, reps if not cold else 60) * 1e3 for _ in range(3)]
    print(f"{tag:52s} {'cold' if cold else 'warm'}  us: " + " ".join(f"{t:7.1f}" for t in ts), flush=True)
for rep in range(3):
    for cold in (False, True):
        run("default (both priorities)", 0, cold, 150)
        run("no load-phase priority (512)", 512, cold, 150)
        run("no long-row priority (2048)", 2048, cold, 150)
        run("no priorities (512+2048)", 512 + 2048, cold, 150)
        run("long-row rule: waves 6, 7 (8192)", 8192, cold, 150)
