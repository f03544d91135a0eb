"""One-off: time the CPU oracle's hot path (stencils + restriction + assembly, ONE thread: it is a restatement for
checking, not a tuned CPU code) and its PCG (all host threads) on the headline inputs, next to the device numbers.
Inputs come from the device pre-pass (bit-identical to the oracle's own, tests/test_gpu_prepass.py)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch

from adaptiveviscositysolver_amd import DevicePrepass, capi, scenes
from oracle import oracle as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
levels = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda:0")
sc = scenes.fat_beam(n, levels, device=dev)
pp = DevicePrepass(sc.res, sc.dx, sc.levels)
pi = pp.run(sc.liquid, sc.solid)
o = O.Oracle(*sc.res, sc.dx, sc.dt, pi.levels, True)
t0 = time.time()
o.set_levels(pi.levels)
o.set_field(O.F_LIQUID, sc.liquid.cpu().numpy())
o.set_field(O.F_VISCOSITY, None, float(sc.viscosity))
o.set_field(O.F_DENSITY, None, float(sc.density))
o.set_field(O.F_CENTERW, pp.weights(capi.FIELD_CENTER_WEIGHTS))
for a in range(3):
    o.set_field(O.F_VELOCITY + a, sc.velocity[a].cpu().numpy())
    o.set_field(O.F_EDGEW + a, pp.weights(capi.FIELD_EDGE_WEIGHTS, a))
    o.set_field(O.F_FACEW + a, pp.weights(capi.FIELD_FACE_WEIGHTS, a))
for l in range(pi.levels):
    o.set_labels(l, pp.labels(l))
    for a in range(3):
        o.set_index(O.I_VELOCITY, l, a, pp.index(capi.INDEX_VELOCITY, l, a))
        o.set_index(O.I_EDGE, l, a, pp.index(capi.INDEX_EDGE, l, a))
    o.set_index(O.I_CENTER, l, 0, pp.index(capi.INDEX_CENTER, l))
o.finalize_indices()
pp.close()
print(f"inputs to the oracle: {time.time()-t0:.1f} s (download + copy)", flush=True)
t = time.time(); o.build_stencils(); t_st = time.time() - t
t = time.time(); o.build_initial_guess(); t_ig = time.time() - t
t = time.time(); o.assemble(); t_as = time.time() - t
c = o.csr()
print(f"CPU oracle hot path at {n}^3/{pi.levels} levels, n={c.n} nnz={len(c.col)}: stencils {t_st:.2f} s, "
      f"initial guess {t_ig:.2f} s, assembly {t_as:.2f} s (1 thread)", flush=True)
threads = O.max_threads()
x, info = o.solve(1e-3, 2500, threads)
print(f"CPU oracle PCG: {info.iterations} iterations in {info.seconds:.2f} s on {threads} threads "
      f"= {info.iterations/info.seconds:.1f} it/s", flush=True)
