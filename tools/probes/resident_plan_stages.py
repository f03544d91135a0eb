"""Where the CU-resident loop's plan spends its time for a NEW matrix in a warmed-up context (AVS_CG_RESIDENT_VERBOSE=1 prints the stages)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["AVS_CG_RESIDENT_VERBOSE"] = "1"
import torch
from adaptiveviscositysolver_amd import DevicePrepass, ViscositySolve, scenes
dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "beam"
sc = {"beam": lambda: scenes.viscous_beam_scene(device=dev), "buckling": lambda: scenes.viscous_buckling_scene(device=dev),
      "c2": lambda: scenes.fat_beam(128, 3, device=dev), "b256": lambda: scenes.fat_beam(256, 4, device=dev)}[which]()
fsc = scenes.crop_to_field(sc)
pp = DevicePrepass(sc.res, sc.dx, sc.levels, field_res=sc.field_res)
pi = pp.run(fsc.liquid, fsc.solid)
s = ViscositySolve(sc.res, sc.dx, sc.dt, pi.levels, field_res=sc.field_res)
pp.apply(s); s.set_scene_fields(fsc)
for k in range(3):
    s.assemble()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    info = s.solve(1e-3, 2500)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    s.solve(1e-3, 2500)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"== cycle {k}: new-matrix solve {(t1 - t0) * 1e3:.2f} ms, repeated solve {(t2 - t1) * 1e3:.2f} ms, iterations {info.iterations}, resident {info.resident}", file=sys.stderr)
