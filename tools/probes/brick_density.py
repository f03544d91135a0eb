"""Rows per tile against brick / stream SpMV time on several scenes (which form should the auto mode pick?)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from adaptiveviscositysolver_amd import DevicePrepass, ViscositySolve, scenes
dev = torch.device("cuda:0")
os.environ["AVS_BRICK"] = "1"
os.environ["AVS_CG_RESIDENT"] = "0"
cases = [("beam", 512, 4, None), ("beam", 256, 4, None), ("sheet", 512, 4, 32), ("sheet", 512, 4, 64), ("sheet", 1024, 5, 32), ("sheet", 1024, 5, 64), ("tank", 256, 4, None), ("tank", 512, 4, None)]
for kind, n, lv, th in cases:
    sc = scenes.fat_beam(n, lv, device=dev) if kind == "beam" else scenes.thin_sheet(n, lv, thickness_cells=th, device=dev) if kind == "sheet" else scenes.tank(n, lv, device=dev)
    pp = DevicePrepass(sc.res, sc.dx, sc.levels); pi = pp.run(sc.liquid, sc.solid)
    s = ViscositySolve(sc.res, sc.dx, sc.dt, pi.levels, probe=True); pp.apply(s); s.set_scene_fields(sc); pp.close()
    del sc; torch.cuda.empty_cache()
    ai = s.assemble()
    f = s.matrix_format()
    if f.brick_tiles <= 0:
        print(kind, n, th, "rows", ai.n_velocity, "no brick form (table", f.value_table_size, ")"); s.close(); continue
    tb = min(s.bench_spmv(100, 50) for _ in range(2)) * 1e3
    ts = min(s.bench_spmv(61, 50) for _ in range(2)) * 1e3
    print(f"{kind} {n} th={th}: rows {ai.n_velocity} tiles {f.brick_tiles} rows/tile {ai.n_velocity / f.brick_tiles:.0f} pattern rows {f.brick_pattern_rows / ai.n_velocity:.3f} "
          f"brick {tb:.1f} us stream {ts:.1f} us ratio {tb / ts:.2f}")
    s.close(); torch.cuda.empty_cache()
