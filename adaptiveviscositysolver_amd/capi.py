"""ctypes binding of the C ABI in include/avs.h (libavs_hip.so, built in-tree for gfx950).

This is plumbing only: it declares the prototypes, turns avs_status into exceptions and offers
small helpers to hand torch / numpy buffers across the boundary.  It fails loudly when the HIP
library is missing -- there is no CPU fallback of any kind in the product path.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AVS_LIB_PATH", os.path.join(_HERE, "libavs_hip.so"))
PROBE_LIB_PATH = os.environ.get("AVS_PROBE_LIB_PATH", os.path.join(_HERE, "libavs_probe.so"))

MAX_LEVELS = 8
EDGE_STENCIL_CAP, CENTER_STENCIL_CAP = 32, 8
EDGE_BOUNDARY_CAP, CENTER_BOUNDARY_CAP = 4, 2
UNIQUE_ID_BYTES = 128
DIST_BLOB_BYTES = 512

OK, EINVAL, ENOMEM, EHIP, ERCCL, EINTERNAL, ESTATE = range(7)
STATUS_NAMES = {0: "AVS_OK", 1: "AVS_EINVAL", 2: "AVS_ENOMEM", 3: "AVS_EHIP", 4: "AVS_ERCCL",
                5: "AVS_EINTERNAL", 6: "AVS_ESTATE"}
MEM_HOST, MEM_DEVICE = 0, 1
PRECISION_F64, PRECISION_F32 = 0, 1   # avs_desc.precision (SolveType of the reference, util.h:25-37)
(OPTION_PRECONDITIONER, OPTION_RESIDENT_LOOP, OPTION_TRANSPORT, OPTION_PARANOID, OPTION_GRAPH_REPLAY, OPTION_BRICK_FORM,
 OPTION_FUSED_SCALAR_STEPS, OPTION_RELOAD_ENVIRONMENT, OPTION_F32_VECTORS, OPTION_FUSED_VECTOR_UPDATE) = range(10)  # avs_set_solver_option
USE_TRANSPORT_AUTO, USE_TRANSPORT_RCCL, USE_TRANSPORT_DIRECT = 0, 1, 2
BRICK_AUTO, BRICK_NEVER, BRICK_ALWAYS, BRICK_TUNE = -1, 0, 1, 2
PRECONDITIONER_JACOBI, PRECONDITIONER_NONE = 0, 1
INACTIVE, ACTIVE, UP, DOWN = 0, 1, 2, 3
UNASSIGNED, SOLIDBOUNDARY, OUTSIDE = -1, -2, -3
INDEX_VELOCITY, INDEX_EDGE, INDEX_CENTER = 0, 1, 2
(FIELD_CENTER_WEIGHTS, FIELD_EDGE_WEIGHTS, FIELD_FACE_WEIGHTS, FIELD_VISCOSITY, FIELD_DENSITY,
 FIELD_VELOCITY, FIELD_SOLID_VELOCITY) = range(7)

# every symbol include/avs.h declares (checked by tests/test_capi_symbols.py)
EXPORTED_SYMBOLS = [
    "avs_last_error", "avs_version", "avs_abi_version", "avs_cancel", "avs_cancel_clear", "avs_create", "avs_destroy", "avs_set_labels",
    "avs_set_index_field", "avs_set_dof_counts", "avs_set_scalar_field", "avs_build_stencils",
    "avs_build_initial_guess", "avs_build_system", "avs_assemble", "avs_solve", "avs_set_solver_option",
    "avs_get_assembly_info", "avs_get_matrix_format", "avs_get_solution", "avs_get_initial_guess", "avs_get_csr",
    "avs_get_edge_stencils", "avs_get_center_stencils", "avs_pcg_csr",
    "avs_prepass_create", "avs_prepass_destroy", "avs_prepass_run",
    "avs_prepass_get_info", "avs_prepass_get_labels", "avs_prepass_get_mask", "avs_prepass_get_index",
    "avs_prepass_get_weights", "avs_prepass_get_regular_index", "avs_prepass_apply", "avs_set_regular_index_field",
    "avs_transfer_to_regular_grid", "avs_transfer_to_regular_grid_in_place", "avs_get_node_grid", "avs_get_dof_table", "avs_plan_owners", "avs_plan_create", "avs_plan_get_sizes",
    "avs_plan_get_arrays", "avs_plan_destroy", "avs_dist_get_unique_id", "avs_dist_init",
    "avs_local_group_create", "avs_local_group_destroy", "avs_dist_init_local", "avs_dist_partition",
    "avs_spmv_tile_rows", "avs_dist_assemble", "avs_dist_get_plan_sizes", "avs_dist_get_overlap_tiles", "avs_dist_get_plan_arrays", "avs_dist_solve", "avs_dist_get_solution",
    "avs_dist_get_info", "avs_dist_init_hosted", "avs_dist_export_blob", "avs_dist_import_blobs",
    "avs_prepass_set_slab", "avs_prepass_get_window", "avs_dist_bind_prepass", "avs_dist_get_cuts", "avs_set_solution",
]
# avs_allreduce_i32_fn: avs_status (*)(int32_t *device_data, int64_t count, void *stream, void *user)
ALLREDUCE_I32_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p)
# include/avs_probe.h: exported by libavs_probe.so only (the -DAVS_PROBES build of the same sources)
PROBE_SYMBOLS = ["avs_spmv_csr", "avs_bench_spmv", "avs_spmv_sell", "avs_bench_stream", "avs_brick_spmv_probe", "avs_spmv_solver_form",
                 "avs_dist_spmv_local_form", "avs_brick_wave_stats"]
_VOID_RETURN = ("avs_last_error", "avs_version", "avs_destroy", "avs_plan_destroy", "avs_local_group_destroy",
                "avs_prepass_destroy")


class AvsError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {message}")
        self.status = status


class Desc(C.Structure):
    _fields_ = [("nx", C.c_int32), ("ny", C.c_int32), ("nz", C.c_int32), ("dx", C.c_double),
                ("dt", C.c_double), ("levels", C.c_int32), ("use_enhanced_gradients", C.c_int32),
                ("device", C.c_int32), ("stream", C.c_void_p),
                ("field_nx", C.c_int32), ("field_ny", C.c_int32), ("field_nz", C.c_int32), ("precision", C.c_int32)]


class SolveInfo(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("converged", C.c_int32), ("error", C.c_double),
                ("rhs_norm2", C.c_double), ("n", C.c_int64), ("nnz", C.c_int64),
                ("solve_ms", C.c_double), ("spmv_ms", C.c_double), ("resident", C.c_int32), ("cancelled", C.c_int32)]


class PlanSizes(C.Structure):
    _fields_ = [("n_own", C.c_int64), ("n_halo", C.c_int64), ("nnz_local", C.c_int64), ("n_send", C.c_int64),
                ("n_peers", C.c_int32)]


class PrepassDesc(C.Structure):
    _fields_ = [("nx", C.c_int32), ("ny", C.c_int32), ("nz", C.c_int32), ("dx", C.c_double),
                ("desired_levels", C.c_int32), ("n_super", C.c_int32), ("extrapolation_scale", C.c_double),
                ("device", C.c_int32), ("stream", C.c_void_p),
                ("field_nx", C.c_int32), ("field_ny", C.c_int32), ("field_nz", C.c_int32)]


class PrepassInfo(C.Structure):
    _fields_ = [("levels", C.c_int32), ("n_velocity", C.c_int64), ("n_edge", C.c_int64), ("n_center", C.c_int64),
                ("n_regular", C.c_int64), ("weights_ms", C.c_double), ("octree_ms", C.c_double), ("classify_ms", C.c_double),
                ("number_ms", C.c_double)]


class AssemblyInfo(C.Structure):
    _fields_ = [("n_velocity", C.c_int64), ("n_edge", C.c_int64), ("n_center", C.c_int64),
                ("nnz", C.c_int64), ("raw_triplets", C.c_int64), ("stencil_ms", C.c_double),
                ("guess_ms", C.c_double), ("system_ms", C.c_double), ("csr_ms", C.c_double)]


class DistInfo(C.Structure):
    _fields_ = [("world_size", C.c_int32), ("rccl_ranks", C.c_int32), ("transport", C.c_int32), ("graph_replay", C.c_int32),
                ("launches_per_iteration", C.c_int32), ("collectives_per_iteration", C.c_int32),
                ("selftest_rounds", C.c_int32), ("paranoid", C.c_int32), ("selftest_bad_entries", C.c_int64)]


class MatrixFormat(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("reordered", C.c_int32), ("value_table_size", C.c_int32), ("column_bits", C.c_int32),
                ("bytes_per_nonzero", C.c_int32), ("tile_local_tables", C.c_int32),
                ("column_windows", C.c_int32), ("brick_tiles", C.c_int32), ("brick_patterns", C.c_int32), ("_pad", C.c_int32),
                ("brick_pattern_rows", C.c_int64), ("brick_bytes", C.c_int64), ("brick_walk", C.c_int32), ("brick_value_codes", C.c_int32),
                ("fused_vector_update", C.c_int32), ("fused_vector_faults", C.c_int32)]

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.struct_size = C.sizeof(MatrixFormat)


def source_fingerprint():
    """sha256 (16 hex digits) over the library's sources: kernels, internal headers, the public header.  Counter records under
    profiles/ carry the fingerprint of the tree they were taken with; bench.py only quotes a record whose fingerprint is the running one."""
    import glob
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
    files = sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.hpp")) + glob.glob(os.path.join(csrc, "*.inl")) +
                   glob.glob(os.path.join(csrc, "*.cpp")) + [os.path.join(csrc, "Makefile")] +
                   glob.glob(os.path.join(os.path.dirname(csrc), "..", "include", "*.h")))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


_lib = None
_probe_lib = None


def load_probe():
    """libavs_probe.so: the product sources built with -DAVS_PROBES (+ the entries of include/avs_probe.h).  Tools and tests only."""
    return load(probe=True)


def load(probe=False):
    """Load libavs_hip.so (or the probe build); raises if it has not been built (python __graft_entry__.py build)."""
    global _lib, _probe_lib
    if probe and _probe_lib is not None:
        return _probe_lib
    if not probe and _lib is not None:
        return _lib
    path = PROBE_LIB_PATH if probe else LIB_PATH
    if not os.path.exists(path):
        raise FileNotFoundError(
            f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    L = C.CDLL(path)
    vp, i32, i64, f64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_double, C.c_float
    L.avs_last_error.restype = C.c_char_p
    L.avs_version.restype = C.c_char_p
    L.avs_create.argtypes = [C.POINTER(Desc), C.POINTER(vp)]
    L.avs_destroy.argtypes = [vp]
    L.avs_destroy.restype = None
    L.avs_set_labels.argtypes = [vp, i32, vp, i32]
    L.avs_set_index_field.argtypes = [vp, i32, i32, i32, vp, i32]
    L.avs_set_dof_counts.argtypes = [vp, i64, i64, i64]
    L.avs_set_scalar_field.argtypes = [vp, i32, i32, vp, f32, i32]
    L.avs_build_stencils.argtypes = [vp]
    L.avs_build_initial_guess.argtypes = [vp]
    L.avs_build_system.argtypes = [vp]
    L.avs_assemble.argtypes = [vp, C.POINTER(AssemblyInfo)]
    L.avs_solve.argtypes = [vp, f64, i32, C.POINTER(SolveInfo)]
    L.avs_cancel.argtypes = [vp]
    L.avs_cancel_clear.argtypes = [vp]
    L.avs_abi_version.restype = C.c_int32
    L.avs_set_solver_option.argtypes = [vp, i32, i32]
    L.avs_get_assembly_info.argtypes = [vp, C.POINTER(AssemblyInfo)]
    L.avs_get_matrix_format.argtypes = [vp, C.POINTER(MatrixFormat)]
    L.avs_get_solution.argtypes = [vp, vp, i64, i32]
    L.avs_get_initial_guess.argtypes = [vp, vp, i64, i32]
    L.avs_set_solution.argtypes = [vp, vp, i64, i32]
    L.avs_get_csr.argtypes = [vp, vp, vp, vp, vp, i32]
    L.avs_get_edge_stencils.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32]
    L.avs_get_center_stencils.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32]
    L.avs_pcg_csr.argtypes = [i64, vp, vp, vp, vp, vp, f64, i32, i32, i32, vp, C.POINTER(SolveInfo)]
    if probe:
        L.avs_spmv_csr.argtypes = [i64, vp, vp, vp, vp, vp, i32, i32, vp]
        L.avs_bench_spmv.argtypes = [vp, i32, i32, C.POINTER(f64)]
        L.avs_spmv_sell.argtypes = [i64, vp, vp, vp, vp, vp, i32, vp, C.POINTER(f64)]
        L.avs_bench_stream.argtypes = [i32, i64, i32, i32, C.POINTER(f64)]
        L.avs_spmv_solver_form.argtypes = [vp, vp, vp, i32, C.POINTER(f64)]
        L.avs_dist_spmv_local_form.argtypes = [vp, vp, vp, i32, C.POINTER(f64)]
        L.avs_brick_wave_stats.argtypes = [vp, C.POINTER(f64)]
    L.avs_prepass_create.argtypes = [C.POINTER(PrepassDesc), C.POINTER(vp)]
    L.avs_prepass_destroy.argtypes = [vp]
    L.avs_prepass_destroy.restype = None
    L.avs_prepass_run.argtypes = [vp, vp, vp, i32]
    L.avs_prepass_get_info.argtypes = [vp, C.POINTER(PrepassInfo)]
    L.avs_prepass_get_labels.argtypes = [vp, i32, vp, i32]
    L.avs_prepass_get_mask.argtypes = [vp, vp, i32]
    L.avs_prepass_get_index.argtypes = [vp, i32, i32, i32, vp, i32]
    L.avs_prepass_get_weights.argtypes = [vp, i32, i32, vp, i32]
    L.avs_prepass_apply.argtypes = [vp, vp]
    L.avs_prepass_get_regular_index.argtypes = [vp, i32, vp, i32]
    L.avs_set_regular_index_field.argtypes = [vp, i32, vp, i32]
    L.avs_transfer_to_regular_grid.argtypes = [vp, vp, vp, vp, i32]
    L.avs_transfer_to_regular_grid_in_place.argtypes = [vp, vp, vp, vp]
    L.avs_get_node_grid.argtypes = [vp, i32, vp, vp, vp, vp, i32]
    L.avs_get_dof_table.argtypes = [vp, i32, vp, i32]
    L.avs_plan_owners.argtypes = [i64, vp, vp, i32, i32, i32, i32, vp]
    L.avs_plan_create.argtypes = [i64, vp, vp, vp, i32, i32, C.POINTER(vp)]
    L.avs_plan_get_sizes.argtypes = [vp, C.POINTER(PlanSizes)]
    L.avs_plan_get_arrays.argtypes = [vp] + [vp] * 9
    L.avs_plan_destroy.argtypes = [vp]
    L.avs_plan_destroy.restype = None
    L.avs_dist_get_unique_id.argtypes = [vp]
    L.avs_dist_init.argtypes = [vp, vp, i32, i32]
    L.avs_local_group_create.argtypes = [i32, C.POINTER(vp)]
    L.avs_local_group_destroy.argtypes = [vp]
    L.avs_local_group_destroy.restype = None
    L.avs_dist_init_local.argtypes = [vp, vp, i32]
    L.avs_dist_partition.argtypes = [vp, i32]
    L.avs_dist_get_plan_sizes.argtypes = [vp, C.POINTER(PlanSizes)]
    L.avs_dist_assemble.argtypes = [vp, i32, C.POINTER(AssemblyInfo)]
    L.avs_spmv_tile_rows.argtypes = []
    L.avs_spmv_tile_rows.restype = i32
    L.avs_dist_get_overlap_tiles.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
    L.avs_dist_get_plan_arrays.argtypes = [vp] + [vp] * 9
    L.avs_dist_solve.argtypes = [vp, f64, i32, C.POINTER(SolveInfo)]
    L.avs_dist_get_solution.argtypes = [vp, vp, i64, i32]
    L.avs_dist_get_info.argtypes = [vp, C.POINTER(DistInfo)]
    L.avs_dist_init_hosted.argtypes = [vp, i32, i32]
    L.avs_dist_export_blob.argtypes = [vp, vp]
    L.avs_dist_import_blobs.argtypes = [vp, vp]
    L.avs_prepass_set_slab.argtypes = [vp, i32, vp, i32, i32, ALLREDUCE_I32_FN, vp]
    L.avs_prepass_get_window.argtypes = [vp, vp, vp, vp]
    L.avs_dist_bind_prepass.argtypes = [vp, vp, i32, vp]
    L.avs_dist_get_cuts.argtypes = [vp, i32, C.POINTER(i32), vp]
    for name in EXPORTED_SYMBOLS + (PROBE_SYMBOLS if probe else []):
        fn = getattr(L, name)
        if name not in _VOID_RETURN:
            fn.restype = C.c_int
    if probe:
        _probe_lib = L
    else:
        _lib = L
    return L


def check(status):
    if status != OK:
        msgs = [m for m in (l.avs_last_error().decode("utf-8", "replace") for l in (_lib, _probe_lib) if l is not None) if m]
        raise AvsError(status, " | ".join(dict.fromkeys(msgs)) or "(no message)")


def ptr_of(buf):
    """(pointer, memspace) of a numpy array or a torch tensor (contiguous)."""
    if buf is None:
        return None, MEM_HOST
    if isinstance(buf, np.ndarray):
        if not buf.flags["C_CONTIGUOUS"]:
            raise ValueError("numpy buffer must be C contiguous")
        return buf.ctypes.data, MEM_HOST
    # torch tensor
    if not buf.is_contiguous():
        raise ValueError("tensor must be contiguous")
    return buf.data_ptr(), (MEM_DEVICE if buf.is_cuda else MEM_HOST)


# --------------------------------------------------------------------------------------------
# host-side partition planner (no GPU needed)
# --------------------------------------------------------------------------------------------
def plan_owners(dof_table, row_ptr, levels, cut_axis, extent_fine, world_size):
    """owner rank per velocity DOF (spatial slabs balanced by nnz)."""
    L = load()
    tab = np.ascontiguousarray(dof_table, np.int32)
    rp = np.ascontiguousarray(row_ptr, np.int32)
    n = len(rp) - 1
    owner = np.empty(n, np.int32)
    check(L.avs_plan_owners(n, tab.ctypes.data, rp.ctypes.data, levels, cut_axis, extent_fine, world_size,
                            owner.ctypes.data))
    return owner


def plan_create(row_ptr, col, owner, rank, world_size):
    """dict with the local structures of `rank` (see avs_plan_get_arrays)."""
    L = load()
    rp = np.ascontiguousarray(row_ptr, np.int32)
    cl = np.ascontiguousarray(col, np.int32)
    ow = np.ascontiguousarray(owner, np.int32)
    h = C.c_void_p()
    check(L.avs_plan_create(len(rp) - 1, rp.ctypes.data, cl.ctypes.data, ow.ctypes.data, rank, world_size, C.byref(h)))
    try:
        sz = PlanSizes()
        check(L.avs_plan_get_sizes(h, C.byref(sz)))
        a = dict(own_global=np.empty(sz.n_own, np.int32), halo_global=np.empty(sz.n_halo, np.int32),
                 row_ptr_local=np.empty(sz.n_own + 1, np.int32), col_local=np.empty(sz.nnz_local, np.int32),
                 val_src=np.empty(sz.nnz_local, np.int32), peers=np.empty(sz.n_peers, np.int32),
                 send_counts=np.empty(sz.n_peers, np.int32), recv_counts=np.empty(sz.n_peers, np.int32),
                 send_idx=np.empty(sz.n_send, np.int32))
        check(L.avs_plan_get_arrays(h, *[a[k].ctypes.data for k in
                                         ("own_global", "halo_global", "row_ptr_local", "col_local", "val_src",
                                          "peers", "send_counts", "recv_counts", "send_idx")]))
    finally:
        L.avs_plan_destroy(h)
    return a
