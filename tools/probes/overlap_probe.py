"""Can a bandwidth-bound vector kernel hide under the (issue-bound) brick SpMV?  The solver's fused-dot SpMV on the context's stream in a
host thread, a stream of x += a p (3 n doubles of traffic per launch) on a second stream, alone and together.
  python tools/probes/overlap_probe.py [--n 512]"""
import argparse, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from adaptiveviscositysolver_amd import DevicePrepass, ViscositySolve, scenes

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=512)
ap.add_argument("--repeats", type=int, default=200)
a = ap.parse_args()
dev = torch.device("cuda:0")
sc = scenes.fat_beam(a.n, 4, device=dev)
pp = DevicePrepass(sc.res, sc.dx, sc.levels)
pi = pp.run(sc.liquid, sc.solid)
s = ViscositySolve(sc.res, sc.dx, sc.dt, pi.levels, probe=True)
pp.apply(s); s.set_scene_fields(sc); pp.close()
ai = s.assemble()
n = int(ai.n_velocity)
x = torch.zeros(n, dtype=torch.float64, device=dev)
p = torch.ones(n, dtype=torch.float64, device=dev)
side = torch.cuda.Stream()

def spmv(rep):
    return s.bench_spmv(variant=100, repeats=rep)   # ms per launch, synchronises

def axpy(rep):
    with torch.cuda.stream(side):
        for _ in range(rep):
            x.add_(p, alpha=1e-9)
    side.synchronize()

spmv(20); axpy(20)
t0 = time.perf_counter(); ms_spmv = spmv(a.repeats); w_spmv = time.perf_counter() - t0
t0 = time.perf_counter(); axpy(a.repeats); w_axpy = time.perf_counter() - t0
res = {}
th = threading.Thread(target=lambda: res.__setitem__("spmv_ms", spmv(a.repeats)))
t0 = time.perf_counter()
th.start(); axpy(a.repeats); th.join()
w_both = time.perf_counter() - t0
print(f"n = {n}: SpMV alone {ms_spmv * 1e3:.1f} us per launch (wall {w_spmv / a.repeats * 1e6:.1f}), axpy alone {w_axpy / a.repeats * 1e6:.1f} us per launch, "
      f"together: wall {w_both / a.repeats * 1e6:.1f} us per pair, SpMV {res['spmv_ms'] * 1e3:.1f} us per launch while the axpy stream runs")
