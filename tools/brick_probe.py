"""Brick-structured SpMV: build the form with the torch reference builder (tools/brick_build.py) on the GPU, run csrc/avs_brick.hip
through avs_brick_spmv_probe, check y bit for bit against the plain CSR kernel and time it next to the library's default kernel.

  python tools/brick_probe.py --n 512 --levels 4 [--scene beam|sphere|varvisc|sheet] [--out profiles/...json]
"""
import argparse, ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from adaptiveviscositysolver_amd import DevicePrepass, ViscositySolve, capi, scenes
import brick_build as bb

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=128)
ap.add_argument("--levels", type=int, default=4)
ap.add_argument("--scene", default="beam")
ap.add_argument("--repeats", type=int, default=50)
ap.add_argument("--out", default="")
a = ap.parse_args()
dev = torch.device("cuda:0")
if a.scene == "beam": sc = scenes.fat_beam(a.n, a.levels, device=dev)
elif a.scene == "sphere": sc = scenes.sphere(a.n, a.levels, device=dev)
elif a.scene == "varvisc": sc = scenes.fat_beam(a.n, a.levels, variable_viscosity=True, device=dev)
elif a.scene == "hipbeam": sc = scenes.viscous_beam_scene(dev, 1)
elif a.scene == "hipbuckling": sc = scenes.viscous_buckling_scene(dev, 1)
else: sc = scenes.thin_sheet(a.n, a.levels, thickness_cells=32, device=dev)
fres = getattr(sc, "field_res", None)
pp = DevicePrepass(sc.res, sc.dx, sc.levels, field_res=fres)
if fres is not None: sc = scenes.crop_to_field(sc)
pi = pp.run(sc.liquid, sc.solid)
s = ViscositySolve(sc.res, sc.dx, sc.dt, pi.levels, field_res=fres, probe=True)
pp.apply(s); s.set_scene_fields(sc); pp.close()
res = sc.res
del sc; torch.cuda.empty_cache()
ai = s.assemble()
n, nnz = int(ai.n_velocity), int(ai.nnz)
rp = torch.empty(n + 1, dtype=torch.int32, device=dev); col = torch.empty(nnz, dtype=torch.int32, device=dev); val = torch.empty(nnz, dtype=torch.float64, device=dev)
capi.check(s.lib.avs_get_csr(s.h, rp.data_ptr(), col.data_ptr(), val.data_ptr(), None, capi.MEM_DEVICE))
tab = torch.empty((n, 4), dtype=torch.int32, device=dev)
capi.check(s.lib.avs_get_dof_table(s.h, capi.INDEX_VELOCITY, tab.data_ptr(), capi.MEM_DEVICE))
ms_default = s.bench_spmv(variant=100, repeats=a.repeats)   # the solver's fused-dot kernel (checks itself against plain CSR)
fmt = s.matrix_format()
L = s.lib
s.close()
torch.cuda.synchronize()
t0 = time.time()
perm, geo = bb.brick_major(tab, res)
rp2, col2, val2 = bb.permute_csr(rp.long(), col.long(), val, perm)
del rp, col, val, tab
table, code = torch.unique(val2, return_inverse=True)
col_bits = max(1, (n - 1).bit_length())
ok_bits = col_bits + max(1, (len(table) - 1).bit_length()) <= 32
form = bb.build(rp2, col2, code, geo, len(table), col_bits)
torch.cuda.synchronize()
build_s = time.time() - t0
st = form["stats"]

class Arr(C.Structure):
    _fields_ = [("ntiles", C.c_int32)] + [(k, C.c_void_p) for k in ("tile_blk", "blocks", "rdesc", "ownslot", "pwords", "sdesc", "swords", "table")] + \
               [("table_size", C.c_int32), ("col_bits", C.c_int32)]
arr = Arr(form["ntiles"], *[form[k].data_ptr() for k in ("tile_blk", "blocks", "rdesc", "ownslot", "pwords", "sdesc", "swords")], table.data_ptr(),
          len(table), col_bits)
L.avs_brick_spmv_probe.argtypes = [C.POINTER(Arr), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.POINTER(C.c_double)]
g = torch.Generator(device=dev); g.manual_seed(7)
x = torch.randn(n, dtype=torch.float64, device=dev, generator=g)
y = torch.full((n,), float("nan"), dtype=torch.float64, device=dev)
partial = torch.zeros(form["ntiles"] * 8 + 16, dtype=torch.float64, device=dev)
ms = C.c_double()
out = {"scene": a.scene, "n": a.n, "rows": n, "nnz": nnz, "values": int(len(table)), "default_kernel_us": ms_default * 1e3,
       "default_format_bytes_per_nnz": int(fmt.bytes_per_nonzero), "torch_build_s": build_s, "form": st, "bits_ok": ok_bits}
if ok_bits:
    capi.check(L.avs_brick_spmv_probe(C.byref(arr), x.data_ptr(), y.data_ptr(), None, a.repeats, None, C.byref(ms)))
    out["brick_us_plain"] = ms.value * 1e3
    capi.check(L.avs_brick_spmv_probe(C.byref(arr), x.data_ptr(), y.data_ptr(), partial.data_ptr(), a.repeats, None, C.byref(ms)))
    out["brick_us_dot"] = ms.value * 1e3
    torch.cuda.synchronize()
    yref = torch.empty_like(y)
    capi.check(L.avs_spmv_csr(n, rp2.to(torch.int32).data_ptr(), col2.to(torch.int32).data_ptr(), val2.data_ptr(), x.data_ptr(), yref.data_ptr(), 14, 1, None))
    torch.cuda.synchronize()
    diff = int((y.view(torch.int64) != yref.view(torch.int64)).sum())
    out["rows_differing_from_plain_csr"] = diff
    out["dot_rel_err"] = float(abs(partial[: form["ntiles"] * 8].sum() - (x * yref).sum()) / abs((x * yref).sum()))
    stored = st["bytes"]["total"] + 16 * n
    out["stored_bytes"] = stored
    out["stored_TBps"] = stored / (out["brick_us_dot"] * 1e-6) / 1e12
    out["survey_8d_TBps"] = (12 * nnz + 4 * (n + 1) + 16 * n) / (out["brick_us_dot"] * 1e-6) / 1e12
print(json.dumps(out, indent=1))
if a.out:
    with open(a.out, "w") as f: json.dump(out, f, indent=1)
