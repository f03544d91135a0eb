"""The brick form on a walled scene (tank: liquid on the domain border, collision SDF on the walls): how many rows are patterns, how many
tiles are E tiles, what the SpMV costs next to the word stream.  python tools/probes/tank_form.py [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from adaptiveviscositysolver_amd import DevicePrepass, ViscositySolve, scenes
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
for name, make in (("tank", lambda: scenes.tank(n, 4, device=dev)), ("beam", lambda: scenes.fat_beam(n, 4, device=dev))):
    sc = make()
    pp = DevicePrepass(sc.res, sc.dx, sc.levels)
    pi = pp.run(sc.liquid, sc.solid)
    s = ViscositySolve(sc.res, sc.dx, sc.dt, pi.levels, probe=True)
    pp.apply(s); s.set_scene_fields(sc); pp.close()
    ai = s.assemble()
    f = s.matrix_format()
    fields = {k: getattr(f, k) for k, _ in f._fields_ if k != "struct_size"}
    ms_b = s.bench_spmv(variant=100, repeats=50)
    ms_s = s.bench_spmv(variant=161, repeats=50)
    print(name, "rows", int(ai.n_velocity), "nnz", int(ai.nnz), fields, f"brick {ms_b * 1e3:.1f} us, stream {ms_s * 1e3:.1f} us", flush=True)
    s.close(); del sc; torch.cuda.empty_cache()
