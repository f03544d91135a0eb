"""Round 6: where does the brick form beat the word stream now?  Per scene: rows per tile (the structural rule's input) and the solve rate with
AVS_OPTION_BRICK_FORM never / always (launch-per-phase loop), one process, one box."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from adaptiveviscositysolver_amd import DevicePrepass, ViscositySolve, scenes, capi
dev = torch.device("cuda:0")
CASES = {
    "beam256_mu": lambda: scenes.fat_beam(256, 4, variable_viscosity=True, device=dev),
    "beam256": lambda: scenes.fat_beam(256, 4, device=dev),
    "sheet512": lambda: scenes.thin_sheet(512, 4, thickness_cells=32, device=dev),
    "sheet512_thin": lambda: scenes.thin_sheet(512, 4, thickness_cells=16, device=dev),
    "tank256": lambda: scenes.tank(256, 4, device=dev),
    "sphere256": lambda: scenes.sphere(256, 4, device=dev),
    "sheet1024": lambda: scenes.thin_sheet(1024, 5, thickness_cells=32, device=dev),
    "sphere512": lambda: scenes.sphere(512, 4, device=dev),
    "sheet512_8": lambda: scenes.thin_sheet(512, 4, thickness_cells=8, device=dev),
    "sheet512_4": lambda: scenes.thin_sheet(512, 3, thickness_cells=4, device=dev),
    "beam512_mu": lambda: scenes.fat_beam(512, 4, variable_viscosity=True, device=dev),
    "tank512": lambda: scenes.tank(512, 4, device=dev),
}
for name in (sys.argv[1:] or list(CASES)):
    sc = CASES[name]()
    pp = DevicePrepass(sc.res, sc.dx, sc.levels); pi = pp.run(sc.liquid, sc.solid)
    s = ViscositySolve(sc.res, sc.dx, sc.dt, pi.levels); pp.apply(s); s.set_scene_fields(sc); pp.close()
    s.set_solver_option(capi.OPTION_RESIDENT_LOOP, 0)
    out = []
    for mode in (0, 1, 0, 1):
        s.set_solver_option(capi.OPTION_BRICK_FORM, mode)
        ai = s.assemble()
        f = s.matrix_format()
        s.solve(1e-3, 3000)
        best = 0.
        for _ in range(2):
            info = s.solve(1e-3, 3000)
            best = max(best, info.iterations / info.solve_ms * 1e3)
        out.append((mode, int(f.brick_tiles), best, info.spmv_ms * 1e3, info.iterations))
        if mode: extra = f"pattern rows {int(f.brick_pattern_rows) / int(ai.n_velocity):.3f} of n, {int(f.brick_patterns)} patterns, form {int(f.brick_bytes) / int(ai.nnz):.2f} B/nnz"
        else: extra0 = f"stream {int(f.bytes_per_nonzero)} B/nnz, table {int(f.value_table_size)}, tile tables {int(f.tile_local_tables)}"
    n = int(ai.n_velocity)
    tiles = max(o[1] for o in out)
    print(f"{name:16s} n={n:9d} tiles={tiles:6d} rows/tile={n / max(1, tiles):6.1f} value_codes={int(f.brick_value_codes)} | " +
          " | ".join(f"brick={m}: {r:8.0f} it/s, SpMV {u:6.1f} us, {it} it" for m, t, r, u, it in out[:2]) + " | " + extra + " | " + extra0, flush=True)
    s.close()
    del sc
    torch.cuda.empty_cache()
