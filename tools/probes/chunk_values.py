"""Distinct matrix values per chunk of consecutive rows (brick-major order) for a variable-viscosity system: would a per-workgroup
dictionary of the CU-resident loop fit 11 bits?"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from adaptiveviscositysolver_amd import DevicePrepass, ViscositySolve, capi, scenes
dev = torch.device("cuda:0")
n0 = int(sys.argv[1]) if len(sys.argv) > 1 else 256
sc = scenes.fat_beam(n0, 4, variable_viscosity=True, device=dev)
pp = DevicePrepass(sc.res, sc.dx, sc.levels); pi = pp.run(sc.liquid, sc.solid)
s = ViscositySolve(sc.res, sc.dx, sc.dt, pi.levels); pp.apply(s); s.set_scene_fields(sc); pp.close()
nx, ny, nz = sc.res
del sc; torch.cuda.empty_cache()
ai = s.assemble(); n, nnz = int(ai.n_velocity), int(ai.nnz)
rp = torch.empty(n + 1, dtype=torch.int32, device=dev); col = torch.empty(nnz, dtype=torch.int32, device=dev); val = torch.empty(nnz, dtype=torch.float64, device=dev)
capi.check(s.lib.avs_get_csr(s.h, rp.data_ptr(), col.data_ptr(), val.data_ptr(), None, capi.MEM_DEVICE)); del col
tab = torch.empty((n, 4), dtype=torch.int32, device=dev)
capi.check(s.lib.avs_get_dof_table(s.h, capi.INDEX_VELOCITY, tab.data_ptr(), capi.MEM_DEVICE)); s.close()
lv = (tab[:, 0] & 0xff).long()
P = [(tab[:, 1 + k].long() << lv) for k in range(3)]
P = [P[0].clamp(max=nx - 1), P[1].clamp(max=ny - 1), P[2].clamp(max=nz - 1)]
nbx, nby = (nx + 7) >> 3, (ny + 7) >> 3
key = ((((P[2] >> 3) * nby + (P[1] >> 3)) * nbx + (P[0] >> 3)) << 9) | ((P[2] & 7) << 6) | ((P[1] & 7) << 3) | (P[0] & 7)
perm = torch.sort(key, stable=True).indices
inv = torch.empty_like(perm); inv[perm] = torch.arange(n, device=dev)
lens = (rp[1:] - rp[:-1]).long()
row_new = inv[torch.repeat_interleave(torch.arange(n, device=dev), lens)]
uv, code = torch.unique(val, return_inverse=True)
out = {"rows": n, "nnz": nnz, "distinct_values": int(uv.numel())}
for chunk in (512, 2500, 5000, 10000):
    c = row_new // chunk
    k2 = torch.unique(c * int(uv.numel()) + code)
    per = torch.zeros(int(c.max()) + 1, dtype=torch.int64, device=dev).index_add_(0, k2 // int(uv.numel()), torch.ones_like(k2))
    q = [float(x) for x in torch.quantile(per.double(), torch.tensor([0.1, 0.5, 0.9, 1.0], device=dev, dtype=torch.float64))]
    out[f"chunk_{chunk}"] = {"distinct_per_chunk_p10_p50_p90_max": q}
print(json.dumps(out, indent=1))
