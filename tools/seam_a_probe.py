"""Seam A (avs_pcg_csr: the caller's CSR in the reference numbering) at the headline size."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from adaptiveviscositysolver_amd import DevicePrepass, ViscositySolve, capi, scenes

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda:0")
sc = scenes.fat_beam(n, 4, device=dev)
pp = DevicePrepass(sc.res, sc.dx, sc.levels)
pi = pp.run(sc.liquid, sc.solid)
s = ViscositySolve(sc.res, sc.dx, sc.dt, pi.levels, device=0)
pp.apply(s); pp.close()
s.set_scene_fields(sc)
ai = s.assemble()
nn, nnz = ai.n_velocity, ai.nnz
rp = torch.empty(nn + 1, dtype=torch.int32, device=dev); col = torch.empty(nnz, dtype=torch.int32, device=dev)
val = torch.empty(nnz, dtype=torch.float64, device=dev); rhs = torch.empty(nn, dtype=torch.float64, device=dev)
capi.check(s.lib.avs_get_csr(s.h, rp.data_ptr(), col.data_ptr(), val.data_ptr(), rhs.data_ptr(), capi.MEM_DEVICE))
x0 = torch.empty(nn, dtype=torch.float64, device=dev)
capi.check(s.lib.avs_get_initial_guess(s.h, x0.data_ptr(), nn, capi.MEM_DEVICE))
for rep in range(2):
    x = x0.clone()
    si = capi.SolveInfo()
    capi.check(s.lib.avs_pcg_csr(nn, rp.data_ptr(), col.data_ptr(), val.data_ptr(), rhs.data_ptr(), x.data_ptr(), 1e-3, 2500,
                                 capi.MEM_DEVICE, 0, None, C.byref(si)))
    print(f"seam A: {si.iterations} iterations, {si.solve_ms:.1f} ms -> {si.iterations/si.solve_ms*1e3:.0f} it/s, spmv {si.spmv_ms*1e3:.0f} us", flush=True)
