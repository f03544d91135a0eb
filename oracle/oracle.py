"""ctypes wrapper around the CPU oracle (oracle/libavs_oracle.so).

TEST INFRASTRUCTURE ONLY.  May be imported from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never from the product package.  PARITY UNPINNED: see
oracle/avs_oracle.h.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libavs_oracle.so")

EDGE_CAP, CENTER_CAP, EDGE_BCAP, CENTER_BCAP = 32, 8, 4, 2
INACTIVE, ACTIVE, UP, DOWN = 0, 1, 2, 3
FLUID, UNASSIGNED, SOLIDBOUNDARY, OUTSIDE = 0, -1, -2, -3

F_LIQUID, F_SOLID, F_VISCOSITY, F_DENSITY = 0, 1, 2, 3
F_VELOCITY, F_SOLIDVEL, F_FACEW, F_CENTERW, F_EDGEW = 4, 7, 10, 13, 14
I_VELOCITY, I_EDGE, I_CENTER = 0, 1, 2


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "avs_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


class PcgInfo(C.Structure):
    _fields_ = [("iterations", C.c_int), ("error", C.c_double), ("rhs_norm2", C.c_double),
                ("seconds", C.c_double), ("spmv_seconds", C.c_double), ("threads", C.c_int)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        vp, i32, i64, f64 = C.c_void_p, C.c_int, C.c_int64, C.c_double
        L.orc_create.restype = vp
        L.orc_create.argtypes = [i32, i32, i32, f64, f64, i32, i32]
        L.orc_destroy.argtypes = [vp]
        L.orc_set_field.argtypes = [vp, i32, vp, C.c_float]
        L.orc_get_field.argtypes = [vp, i32, vp]
        L.orc_field_size.restype = i64
        L.orc_field_size.argtypes = [vp, i32]
        L.orc_build_weights.argtypes = [vp, i32, i32]
        L.orc_build_octree.argtypes = [vp, f64]
        L.orc_build_indices.argtypes = [vp, f64]
        L.orc_set_levels.argtypes = [vp, i32]
        L.orc_set_labels.argtypes = [vp, i32, vp]
        L.orc_set_index.argtypes = [vp, i32, i32, i32, vp]
        L.orc_finalize_indices.argtypes = [vp]
        L.orc_levels.argtypes = [vp]
        L.orc_grid_size.restype = i64
        L.orc_grid_size.argtypes = [vp, i32, i32, i32, vp]
        L.orc_get_labels.argtypes = [vp, i32, vp]
        L.orc_get_mask.argtypes = [vp, vp]
        L.orc_get_index.argtypes = [vp, i32, i32, i32, vp]
        L.orc_count.restype = i64
        L.orc_count.argtypes = [vp, i32]
        L.orc_get_dof_table.argtypes = [vp, i32, vp]
        L.orc_build_stencils.argtypes = [vp]
        L.orc_build_initial_guess.argtypes = [vp]
        L.orc_assemble.argtypes = [vp]
        L.orc_get_edge_stencils.argtypes = [vp] + [vp] * 6
        L.orc_get_center_stencils.argtypes = [vp] + [vp] * 6
        L.orc_get_initial_guess.argtypes = [vp, vp]
        L.orc_nnz.restype = i64
        L.orc_nnz.argtypes = [vp]
        L.orc_raw_triplets.restype = i64
        L.orc_raw_triplets.argtypes = [vp]
        L.orc_get_csr.argtypes = [vp, vp, vp, vp, vp]
        L.orc_pcg_csr.argtypes = [i64, vp, vp, vp, vp, vp, f64, i32, i32, C.POINTER(PcgInfo)]
        L.orc_spmv_csr.argtypes = [i64, vp, vp, vp, vp, vp, i32]
        L.orc_solve.argtypes = [vp, f64, i32, i32, vp, C.POINTER(PcgInfo)]
        L.orc_set_precision.argtypes = [vp, i32]
        L.orc_set_precision.restype = i32
        L.orc_set_preconditioner.argtypes = [vp, i32]
        L.orc_set_preconditioner.restype = i32
        L.orc_build_regular_indices.argtypes = [vp, f64]
        L.orc_regular_count.restype = i64
        L.orc_regular_count.argtypes = [vp]
        L.orc_get_regular_index.argtypes = [vp, i32, vp]
        L.orc_set_regular_index.argtypes = [vp, i32, vp]
        L.orc_transfer_to_regular_grid.argtypes = [vp, vp, vp, vp, vp]
        L.orc_get_node_grid.argtypes = [vp, i32, vp, vp, vp, vp]
        L.orc_max_threads.restype = i32
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _chk(rc, what):
    if rc != 0:
        raise RuntimeError(f"oracle: {what} failed with code {rc}")


@dataclass
class Csr:
    n: int
    row_ptr: np.ndarray  # int64 [n+1]
    col: np.ndarray      # int32 [nnz]
    val: np.ndarray      # float64 [nnz]
    rhs: np.ndarray      # float64 [n]

    def to_scipy(self):
        import scipy.sparse as sp
        return sp.csr_matrix((self.val, self.col.astype(np.int64), self.row_ptr.astype(np.int64)), shape=(self.n, self.n))


class Oracle:
    """One viscosity step on the CPU: pre-pass -> stencils -> CSR -> Jacobi-PCG."""

    def __init__(self, nx, ny, nz, dx, dt, levels=4, use_enhanced_gradients=True, f32=False):
        """f32: SolveType = fpreal32 (the reference built with USESINGLEPRECISION, util.h:25-37)."""
        self.L = lib()
        self.h = self.L.orc_create(nx, ny, nz, dx, dt, levels, int(bool(use_enhanced_gradients)))
        if not self.h:
            raise ValueError("oracle: bad descriptor (resolution must be powers of two)")
        if f32:
            _chk(self.L.orc_set_precision(self.h, 1), "set_precision")
        self.res = (nx, ny, nz)
        self.dx, self.dt = dx, dt

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.L.orc_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # ---- fields ---------------------------------------------------------------------
    def set_field(self, kind, data=None, const=0.0):
        if data is None:
            _chk(self.L.orc_set_field(self.h, kind, None, float(const)), "set_field")
            return
        a = np.ascontiguousarray(np.asarray(data, dtype=np.float32)).ravel()
        n = self.L.orc_field_size(self.h, kind)
        if a.size != n:
            raise ValueError(f"field {kind}: expected {n} values, got {a.size}")
        _chk(self.L.orc_set_field(self.h, kind, _p(a), 0.0), "set_field")

    def get_field(self, kind):
        n = self.L.orc_field_size(self.h, kind)
        out = np.empty(n, dtype=np.float32)
        _chk(self.L.orc_get_field(self.h, kind, _p(out)), "get_field")
        return out

    # ---- pre-pass -------------------------------------------------------------------
    def build_weights(self, n_super=3, also_face_weights=True):
        _chk(self.L.orc_build_weights(self.h, n_super, int(also_face_weights)), "build_weights")

    def build_octree(self, extrapolation=0.5):
        _chk(self.L.orc_build_octree(self.h, extrapolation), "build_octree")

    def build_indices(self, extrapolation=0.5):
        _chk(self.L.orc_build_indices(self.h, extrapolation), "build_indices")

    def prepass(self, n_super=3, extrapolation=0.5):
        self.build_weights(n_super, True)
        self.build_octree(extrapolation)
        self.build_indices(extrapolation)

    @property
    def levels(self):
        return self.L.orc_levels(self.h)

    def grid_res(self, kind, level, axis=0):
        r = (C.c_int * 3)()
        self.L.orc_grid_size(self.h, kind, level, axis, r)
        return tuple(r)

    def labels(self, level):
        r = self.grid_res(I_CENTER, level)
        out = np.empty(r[0] * r[1] * r[2], dtype=np.int8)
        _chk(self.L.orc_get_labels(self.h, level, _p(out)), "get_labels")
        return out.reshape(r[2], r[1], r[0])

    def mask(self):
        r = self.grid_res(I_CENTER, 0)
        out = np.empty(r[0] * r[1] * r[2], dtype=np.int8)
        _chk(self.L.orc_get_mask(self.h, _p(out)), "get_mask")
        return out.reshape(r[2], r[1], r[0])

    def index(self, kind, level, axis=0):
        r = self.grid_res(kind, level, axis)
        out = np.empty(r[0] * r[1] * r[2], dtype=np.int32)
        _chk(self.L.orc_get_index(self.h, kind, level, axis, _p(out)), "get_index")
        return out.reshape(r[2], r[1], r[0])

    def count(self, kind):
        return int(self.L.orc_count(self.h, kind))

    def dof_table(self, kind):
        out = np.empty((self.count(kind), 4), dtype=np.int32)
        _chk(self.L.orc_get_dof_table(self.h, kind, _p(out)), "get_dof_table")
        return out

    def set_levels(self, levels):
        _chk(self.L.orc_set_levels(self.h, levels), "set_levels")

    def set_labels(self, level, labels):
        a = np.ascontiguousarray(np.asarray(labels, dtype=np.int8)).ravel()
        _chk(self.L.orc_set_labels(self.h, level, _p(a)), "set_labels")

    def set_index(self, kind, level, axis, idx):
        a = np.ascontiguousarray(np.asarray(idx, dtype=np.int32)).ravel()
        _chk(self.L.orc_set_index(self.h, kind, level, axis, _p(a)), "set_index")

    def finalize_indices(self):
        _chk(self.L.orc_finalize_indices(self.h), "finalize_indices")

    # ---- hot path -------------------------------------------------------------------
    def build_stencils(self):
        _chk(self.L.orc_build_stencils(self.h), "build_stencils")

    def build_initial_guess(self):
        _chk(self.L.orc_build_initial_guess(self.h), "build_initial_guess")

    def assemble(self):
        _chk(self.L.orc_assemble(self.h), "assemble")

    def hot_path(self):
        self.build_stencils()
        self.build_initial_guess()
        self.assemble()

    def edge_stencils(self):
        ne = self.count(I_EDGE)
        cnt = np.empty(ne, np.int32); idx = np.empty((EDGE_CAP, ne), np.int32)
        coef = np.empty((EDGE_CAP, ne), np.float64); bcnt = np.empty(ne, np.int32)
        bval = np.empty((EDGE_BCAP, ne), np.float64); w = np.empty(ne, np.float64)
        _chk(self.L.orc_get_edge_stencils(self.h, _p(cnt), _p(idx), _p(coef), _p(bcnt), _p(bval), _p(w)),
             "get_edge_stencils")
        return dict(cnt=cnt, idx=idx, coef=coef, bcnt=bcnt, bval=bval, weight=w)

    def center_stencils(self):
        nc = self.count(I_CENTER)
        n3 = 3 * nc
        cnt = np.empty(n3, np.int32); idx = np.empty((CENTER_CAP, n3), np.int32)
        coef = np.empty((CENTER_CAP, n3), np.float64); bcnt = np.empty(n3, np.int32)
        bval = np.empty((CENTER_BCAP, n3), np.float64); w = np.empty(nc, np.float64)
        _chk(self.L.orc_get_center_stencils(self.h, _p(cnt), _p(idx), _p(coef), _p(bcnt), _p(bval), _p(w)),
             "get_center_stencils")
        return dict(cnt=cnt, idx=idx, coef=coef, bcnt=bcnt, bval=bval, weight=w)

    def initial_guess(self):
        out = np.empty(self.count(I_VELOCITY), np.float64)
        _chk(self.L.orc_get_initial_guess(self.h, _p(out)), "get_initial_guess")
        return out

    def csr(self) -> Csr:
        n = self.count(I_VELOCITY)
        nnz = int(self.L.orc_nnz(self.h))
        rp = np.empty(n + 1, np.int64); col = np.empty(nnz, np.int32)
        val = np.empty(nnz, np.float64); rhs = np.empty(n, np.float64)
        _chk(self.L.orc_get_csr(self.h, _p(rp), _p(col), _p(val), _p(rhs)), "get_csr")
        return Csr(n, rp, col, val, rhs)

    @property
    def raw_triplets(self):
        return int(self.L.orc_raw_triplets(self.h))

    # ---- post-solve transfer ---------------------------------------------------------
    def build_regular_indices(self, extrapolation=0.5):
        _chk(self.L.orc_build_regular_indices(self.h, extrapolation), "build_regular_indices")

    @property
    def regular_count(self):
        return int(self.L.orc_regular_count(self.h))

    def regular_index(self, axis):
        r = self.grid_res(I_VELOCITY, 0, axis)
        out = np.empty(r[0] * r[1] * r[2], dtype=np.int32)
        _chk(self.L.orc_get_regular_index(self.h, axis, _p(out)), "get_regular_index")
        return out.reshape(r[2], r[1], r[0])

    def set_regular_index(self, axis, idx):
        a = np.ascontiguousarray(np.asarray(idx, dtype=np.int32)).ravel()
        _chk(self.L.orc_set_regular_index(self.h, axis, _p(a)), "set_regular_index")

    def transfer_to_regular_grid(self, solution):
        x = np.ascontiguousarray(solution, dtype=np.float64)
        outs = []
        for a in range(3):
            r = self.grid_res(I_VELOCITY, 0, a)
            outs.append(np.empty((r[2], r[1], r[0]), dtype=np.float32))
        _chk(self.L.orc_transfer_to_regular_grid(self.h, _p(x), _p(outs[0]), _p(outs[1]), _p(outs[2])), "transfer")
        return outs

    def node_grid(self, level):
        r = self.grid_res(I_CENTER, level)
        shp = (r[2] + 1, r[1] + 1, r[0] + 1)
        lab = np.empty(shp, np.int8)
        v = [np.empty(shp, np.float32) for _ in range(3)]
        _chk(self.L.orc_get_node_grid(self.h, level, _p(lab), _p(v[0]), _p(v[1]), _p(v[2])), "get_node_grid")
        return lab, v

    def solve(self, tol=1e-3, max_iters=2500, threads=1):
        x = np.empty(self.count(I_VELOCITY), np.float64)
        info = PcgInfo()
        _chk(self.L.orc_solve(self.h, tol, max_iters, threads, _p(x), C.byref(info)), "solve")
        return x, info


def pcg_csr(row_ptr, col, val, b, x0, tol=1e-3, max_iters=2500, threads=1):
    """Jacobi-PCG in Eigen's operation order on an arbitrary CSR system."""
    L = lib()
    rp = np.ascontiguousarray(row_ptr, dtype=np.int64)
    cl = np.ascontiguousarray(col, dtype=np.int32)
    vl = np.ascontiguousarray(val, dtype=np.float64)
    bb = np.ascontiguousarray(b, dtype=np.float64)
    x = np.array(x0, dtype=np.float64, copy=True)
    info = PcgInfo()
    _chk(L.orc_pcg_csr(len(bb), _p(rp), _p(cl), _p(vl), _p(bb), _p(x), tol, max_iters, threads,
                       C.byref(info)), "pcg_csr")
    return x, info


def pcg_csr_ex(row_ptr, col, val, b, x0, tol=1e-3, max_iters=2500, spmv_threads=1, vec_threads=1):
    """Same loop with separate thread counts: (T, 1) = row-parallel SpMV, SERIAL dot products and AXPYs -- the order of additions in every
    dot product is then independent of the thread count (what Eigen's ConjugateGradient does under OpenMP: only the product is threaded)."""
    L = lib()
    L.orc_pcg_csr_ex.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_int, C.c_int,
                                 C.POINTER(PcgInfo)]
    rp = np.ascontiguousarray(row_ptr, dtype=np.int64)
    cl = np.ascontiguousarray(col, dtype=np.int32)
    vl = np.ascontiguousarray(val, dtype=np.float64)
    bb = np.ascontiguousarray(b, dtype=np.float64)
    x = np.array(x0, dtype=np.float64, copy=True)
    info = PcgInfo()
    _chk(L.orc_pcg_csr_ex(len(bb), _p(rp), _p(cl), _p(vl), _p(bb), _p(x), tol, max_iters, spmv_threads, vec_threads, C.byref(info)), "pcg_csr_ex")
    return x, info


def spmv_csr(row_ptr, col, val, x, threads=1):
    L = lib()
    rp = np.ascontiguousarray(row_ptr, dtype=np.int64)
    cl = np.ascontiguousarray(col, dtype=np.int32)
    vl = np.ascontiguousarray(val, dtype=np.float64)
    xx = np.ascontiguousarray(x, dtype=np.float64)
    n = len(rp) - 1                      # rows; x may be longer ([owned | halo] in the partitioned case)
    y = np.empty(n, dtype=np.float64)
    _chk(L.orc_spmv_csr(n, _p(rp), _p(cl), _p(vl), _p(xx), _p(y), threads), "spmv_csr")
    return y


def max_threads():
    return int(lib().orc_max_threads())
