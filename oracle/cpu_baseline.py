#!/usr/bin/env python
"""cpu_baseline.py -- the CPU oracle's Jacobi-PCG timed on the host cores of this box.

TEST INFRASTRUCTURE: executed only by bench.py's `cpu_baseline` leg, in a CLEAN subprocess (no torch,
no second OpenMP runtime in the address space).  The parent sets OMP_PROC_BIND / OMP_PLACES /
OMP_NUM_THREADS in the child's environment BEFORE the interpreter starts, so libgomp sees them.

SURVEY.md 8(d) asks for two variants of the reference's CPU solve on the same CSR system:
  eigen_faithful : OpenMP row-parallel SpMV, serial dots / AXPYs -- what Eigen::ConjugateGradient<..,
                   Lower|Upper> does when built with -fopenmp (reference CMakeLists.txt:27-32)
  all_parallel   : every vector operation parallel too (a generous baseline)

It also times the oracle's own (OpenMP row-parallel) assembly, SURVEY 8(d) "CPU assembly baseline", on a SMALLER instance
of the same scene (asm_*.npy in the same directory: liquid SDF + velocity of the fat beam at 256^3, 4 levels -- the 512^3
pyramids are ~8 GB of host arrays and minutes of serial pre-pass, too much for a default bench run): pre-pass, then
stencils + restriction + CSR, reported as rows per second.

usage: cpu_baseline.py <dir with row_ptr.npy col.npy val.npy rhs.npy x0.npy [asm_meta.json asm_*.npy]> <tol> <budget_seconds> <threads>
prints one JSON object.
"""
from __future__ import annotations

import ctypes as C
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


class PcgInfo(C.Structure):
    _fields_ = [("iterations", C.c_int), ("error", C.c_double), ("rhs_norm2", C.c_double),
                ("seconds", C.c_double), ("spmv_seconds", C.c_double), ("threads", C.c_int)]


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def physical_cores(cpus):
    """distinct (package, core) pairs among the logical CPUs of the affinity mask"""
    seen = set()
    for c in cpus:
        try:
            base = f"/sys/devices/system/cpu/cpu{c}/topology/"
            seen.add((open(base + "physical_package_id").read().strip(), open(base + "core_id").read().strip()))
        except OSError:
            seen.add(("?", c))
    return len(seen)


def main():
    d, tol, budget, threads = sys.argv[1], float(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4])
    L = C.CDLL(os.path.join(HERE, "libavs_oracle.so"))
    vp = C.c_void_p
    L.orc_pcg_csr_ex.argtypes = [C.c_int64, vp, vp, vp, vp, vp, C.c_double, C.c_int, C.c_int, C.c_int, C.POINTER(PcgInfo)]
    rp = np.load(os.path.join(d, "row_ptr.npy")).astype(np.int64)
    col = np.load(os.path.join(d, "col.npy"))
    val = np.load(os.path.join(d, "val.npy"))
    rhs = np.load(os.path.join(d, "rhs.npy"))
    x0 = np.load(os.path.join(d, "x0.npy"))
    n, nnz = len(rhs), int(rp[-1])
    spmv_bytes = 12 * nnz + 4 * (n + 1) + 16 * n      # SURVEY 8(d), same figure as the GPU roofline

    def run(spmv_threads, vec_threads, iters):
        x = x0.copy()
        info = PcgInfo()
        rc = L.orc_pcg_csr_ex(n, rp.ctypes.data, col.ctypes.data, val.ctypes.data, rhs.ctypes.data, x.ctypes.data,
                              tol, iters, spmv_threads, vec_threads, C.byref(info))
        if rc != 0:
            raise RuntimeError(f"orc_pcg_csr_ex failed: {rc}")
        return info

    out = {"cpu_model": cpu_model(), "threads": threads, "n": n, "nnz": nnz, "variants": {}}
    share = budget / 2.0
    WARM = 24   # dropped: thread start-up, first touch of the vectors, cold caches (round-2 review, weak #8: the 6 s samples that
                # included them read 54 it/s where the same loop did 78 it/s over a whole solve)
    for name, vt in (("eigen_faithful", 1), ("all_parallel", threads)):
        run(threads, vt, 4)
        probe = run(threads, vt, WARM)
        per_iter = max(probe.seconds / max(probe.iterations, 1), 1e-6)
        iters = int(max(3, min(2500, share / per_iter)))
        info = run(threads, vt, iters)
        done = max(info.iterations, 1)
        per_spmv = info.spmv_seconds / done
        out["variants"][name] = {
            "iter_per_s": done / info.seconds, "iterations": done, "seconds": info.seconds,
            "spmv_share": info.spmv_seconds / max(info.seconds, 1e-12),
            "spmv_gbps": spmv_bytes / max(per_spmv, 1e-12) / 1e9, "spmv_threads": threads, "vector_threads": vt,
            "warmup_iterations_dropped": WARM + 4}
    meta_path = os.path.join(d, "asm_meta.json")
    if os.path.exists(meta_path):
        out["assembly"] = time_assembly(d, json.load(open(meta_path)), threads)
    print(json.dumps(out))


def time_assembly(d, meta, threads):
    """the oracle's pre-pass + assembly of the scene dumped by bench.py (OMP_NUM_THREADS is already set for this process)"""
    sys.path.insert(0, HERE)
    import oracle as O
    o = O.Oracle(*meta["res"], meta["dx"], meta["dt"], meta["levels"], True)
    o.set_field(O.F_LIQUID, np.load(os.path.join(d, "asm_liquid.npy")))
    o.set_field(O.F_VISCOSITY, None, float(meta["viscosity"]))
    o.set_field(O.F_DENSITY, None, float(meta["density"]))
    for a in range(3):
        o.set_field(O.F_VELOCITY + a, np.load(os.path.join(d, f"asm_vel{a}.npy")))
    t0 = time.perf_counter()
    o.prepass()
    t1 = time.perf_counter()
    o.build_stencils()
    o.build_initial_guess()
    o.assemble()
    t2 = time.perf_counter()
    m = o.csr()
    n, nnz = len(m.rhs), int(m.row_ptr[-1])
    return {"workload": meta["name"], "rows": n, "nnz": nnz, "levels": int(o.levels), "threads": threads,
            "prepass_seconds": t1 - t0, "assembly_seconds": t2 - t1, "rows_per_s": n / max(t2 - t1, 1e-9)}


if __name__ == "__main__":
    main()
