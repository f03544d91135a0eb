import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
from adaptiveviscositysolver_amd import ViscositySolve, scenes
from util import build_pyramid, feed
os.environ["AVS_BRICK"] = "1"; os.environ["AVS_CG_RESIDENT"] = "0"
sc = scenes.fat_beam(64, 3, variable_viscosity=True)
pyr = build_pyramid(sc)
s = ViscositySolve(sc.res, sc.dx, sc.dt, pyr.levels, device=0, probe=True)
feed(s, pyr); s.set_scene_fields(scenes.to_device(sc, torch.device("cuda:0")))
ai = s.assemble(); fmt = s.matrix_format()
print("tiles", fmt.brick_tiles, "vc", fmt.brick_value_codes, "rows", fmt.brick_pattern_rows, "of", ai.n_velocity, "bytes", fmt.brick_bytes, flush=True)
try:
    print("spmv ms", s.bench_spmv(int(sys.argv[1]) if len(sys.argv) > 1 else 0, 2), flush=True)
except Exception as e:
    print("ERR", str(e)[:300], flush=True)
