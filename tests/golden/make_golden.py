"""Generates the golden fixtures in this directory from the CPU oracle (oracle/avs_oracle.c).

The reference ships no vectors and cannot be built here (PARITY UNPINNED, see oracle/avs_oracle.h),
so these fixtures pin the oracle against regressions and give the GPU tests inputs + expected
outputs that travel to the GPU box.  Re-generate with:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

from adaptiveviscositysolver_amd import scenes  # noqa: E402
from oracle import oracle as O  # noqa: E402
from util import oracle_for_scene  # noqa: E402

CASES = {
    "sphere16_L3": lambda: scenes.sphere(16, 3, radius=0.36),
    "beam32_L2_wall_varvisc": lambda: scenes.fat_beam(32, 2, wall=True, variable_viscosity=True),
    # round 3: density TENSOR (cpp:2759-2766) + spatially varying solid velocity (cpp:1896-1905, 1952-1960) around a solid ball
    "sphere32_obstacle_rho_usolid": lambda: scenes.with_sampled_fields(scenes.sphere_with_obstacle(32, 3)),
}


def scene_from_fixture(g):
    """Scene rebuilt from the INPUT arrays stored in a fixture (no re-generation: transcendental functions
    in the scene generators may differ in the last bit between CPU types)."""
    import torch
    res = tuple(int(v) for v in g["res"])
    visc = g["viscosity"]
    return scenes.Scene(res=res, dx=float(g["dx"]), dt=float(g["dt"]), levels=int(g["desired_levels"]),
                        liquid=torch.from_numpy(np.array(g["liquid"])),
                        solid=(None if g["solid"].size == 0 else torch.from_numpy(np.array(g["solid"]))),
                        viscosity=(float(visc) if visc.ndim == 0 else torch.from_numpy(np.array(visc))),
                        density=(float(g["density"]) if g["density"].ndim == 0 else torch.from_numpy(np.array(g["density"]))),
                        velocity=[torch.from_numpy(np.array(g[f"velocity_{ax}"])) for ax in "xyz"],
                        solid_velocity=([torch.from_numpy(np.array(g[f"solid_velocity_{ax}"])) for ax in "xyz"]
                                        if "solid_velocity_x" in g.files else None))


def build(name, fixture=None):
    sc = CASES[name]() if fixture is None else scene_from_fixture(fixture)
    o = oracle_for_scene(sc)
    o.prepass()
    o.build_regular_indices()
    o.hot_path()
    A = o.csr()
    x, info = o.solve(1e-10, 5000)
    out = o.transfer_to_regular_grid(x)
    d = dict(res=np.array(sc.res), dx=sc.dx, dt=sc.dt, levels=o.levels, desired_levels=sc.levels, counts=np.array([o.count(k) for k in range(3)]),
             liquid=sc.liquid.numpy(), velocity_x=sc.velocity[0].numpy(), velocity_y=sc.velocity[1].numpy(),
             velocity_z=sc.velocity[2].numpy(),
             viscosity=(np.float32(sc.viscosity) if isinstance(sc.viscosity, float) else sc.viscosity.numpy()),
             density=(np.float32(sc.density) if isinstance(sc.density, (int, float)) else sc.density.numpy()),
             solid=(np.zeros(0, np.float32) if sc.solid is None else sc.solid.numpy()),
             centerw=o.get_field(O.F_CENTERW), row_ptr=A.row_ptr.astype(np.int32), col=A.col, val=A.val, rhs=A.rhs,
             x0=o.initial_guess(), x=x, iterations=info.iterations)
    if sc.solid_velocity is not None:
        for a, ax in enumerate("xyz"):
            d[f"solid_velocity_{ax}"] = sc.solid_velocity[a].numpy()
    for a in range(3):
        d[f"ridx{a}"] = o.regular_index(a)
        d[f"out{a}"] = out[a]
        d[f"edgew{a}"] = o.get_field(O.F_EDGEW + a)
        d[f"facew{a}"] = o.get_field(O.F_FACEW + a)
    for l in range(o.levels):
        d[f"labels{l}"] = o.labels(l)
        d[f"cidx{l}"] = o.index(O.I_CENTER, l)
        for a in range(3):
            d[f"vidx{l}_{a}"] = o.index(O.I_VELOCITY, l, a)
            d[f"eidx{l}_{a}"] = o.index(O.I_EDGE, l, a)
    es, cs = o.edge_stencils(), o.center_stencils()
    d.update(e_cnt=es["cnt"], e_weight=es["weight"], c_cnt=cs["cnt"], c_weight=cs["weight"])
    return d


if __name__ == "__main__":
    for name in (sys.argv[1:] or CASES):
        d = build(name)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **d)
        print(name, "n", d["counts"], "nnz", len(d["col"]), "iters", d["iterations"], os.path.getsize(path) // 1024, "KiB")
