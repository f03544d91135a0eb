"""Host-side driver of the hot path: mirrors the phase sequence of
HDK_AdaptiveViscosity::solveGasSubclass between cpp:418 and cpp:653, calling only the C ABI
(include/avs.h).  Buffers are numpy arrays (host) or torch tensors (host or HBM); torch is used
for device memory and streams only.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import capi


class ViscositySolve:
    """One adaptive-viscosity solve on one MI355X.

    Typical use (same order as the reference):
        s = ViscositySolve(res, dx, dt, levels)          # cpp:126-275
        s.set_pyramid(pyramid); s.set_scene_fields(scene) # outputs of cpp:233-416
        info = s.assemble()                               # cpp:418-594, 613-614
        sol = s.solve(tol, max_iters)                     # cpp:618-630
        x = s.solution()                                  # viscositySolution, consumed by cpp:661-707
    """

    def __init__(self, res, dx, dt, levels, use_enhanced_gradients=True, device=0, stream=None, field_res=None, precision=capi.PRECISION_F64,
                 probe=False):
        """`res`: octree (power-of-two) level-0 resolution; `field_res`: resolution of the simulation grid the scalar
        fields live on when HDK_OctreeGrid::init had to pad it (oct.cpp:13-24); None = res.  probe=True: the context lives in
        libavs_probe.so (the same sources + the measurement entries of include/avs_probe.h: bench_spmv ...) -- tools and tests only."""
        self.lib = capi.load(probe=probe)
        self.res = tuple(int(r) for r in res)
        fr = tuple(int(r) for r in field_res) if field_res is not None else (0, 0, 0)
        d = capi.Desc(self.res[0], self.res[1], self.res[2], float(dx), float(dt), int(levels),
                      int(bool(use_enhanced_gradients)), int(device), C.c_void_p(stream or 0), fr[0], fr[1], fr[2], int(precision))
        h = C.c_void_p()
        capi.check(self.lib.avs_create(C.byref(d), C.byref(h)))
        self.h = h
        self.levels = int(levels)
        self.precision = int(precision)
        self.counts = None
        self.field_res = tuple(fr[a] or self.res[a] for a in range(3))

    def close(self):
        if getattr(self, "h", None):
            self.lib.avs_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- inputs ---------------------------------------------------------------------------
    def set_labels(self, level, labels):
        p, where = capi.ptr_of(labels)
        capi.check(self.lib.avs_set_labels(self.h, level, p, where))

    def set_index_field(self, kind, level, axis, idx):
        p, where = capi.ptr_of(idx)
        capi.check(self.lib.avs_set_index_field(self.h, kind, level, axis, p, where))

    def set_dof_counts(self, nv, ne, nc):
        capi.check(self.lib.avs_set_dof_counts(self.h, nv, ne, nc))
        self.counts = (int(nv), int(ne), int(nc))

    def _field_lattice(self, kind, axis, res):
        """(nz, ny, nx) of a level-0 scalar field's lattice on a grid of resolution `res` (x, y, z)."""
        r = list(res)
        if kind in (capi.FIELD_FACE_WEIGHTS, capi.FIELD_VELOCITY, capi.FIELD_SOLID_VELOCITY):
            r[axis] += 1
        elif kind == capi.FIELD_EDGE_WEIGHTS:
            r = [r[b] + (1 if b != axis else 0) for b in range(3)]
        return (r[2], r[1], r[0])

    def set_field(self, kind, axis=0, data=None, const=0.0):
        if data is not None and self.field_res != self.res:
            # the library expects the SIMULATION grid's lattice; crop arrays given on the padded octree lattice
            full, want = self._field_lattice(kind, axis, self.res), self._field_lattice(kind, axis, self.field_res)
            if tuple(data.shape) == full:
                data = data[:want[0], :want[1], :want[2]]
                data = data.contiguous() if hasattr(data, "contiguous") else np.ascontiguousarray(data)
            assert tuple(data.shape) == want, (tuple(data.shape), want)
        p, where = capi.ptr_of(data)
        capi.check(self.lib.avs_set_scalar_field(self.h, kind, axis, p, float(const), where))

    def set_pyramid(self, pyr):
        """Upload a prepass.Pyramid (labels, index pyramids, counts, integration weights)."""
        def c(t, dtype):
            if isinstance(t, np.ndarray):
                return np.ascontiguousarray(t, dtype=dtype)
            import torch
            tdt = {np.int8: torch.int8, np.int32: torch.int32, np.float32: torch.float32}[dtype]
            return t.to(tdt).contiguous()
        for l in range(pyr.levels):
            self.set_labels(l, c(pyr.labels[l], np.int8))
            for a in range(3):
                self.set_index_field(capi.INDEX_VELOCITY, l, a, c(pyr.vidx[l][a], np.int32))
                self.set_index_field(capi.INDEX_EDGE, l, a, c(pyr.eidx[l][a], np.int32))
            self.set_index_field(capi.INDEX_CENTER, l, 0, c(pyr.cidx[l], np.int32))
        self.set_dof_counts(pyr.n_velocity, pyr.n_edge, pyr.n_center)
        self.set_field(capi.FIELD_CENTER_WEIGHTS, 0, c(pyr.center_weights, np.float32))
        for a in range(3):
            self.set_field(capi.FIELD_EDGE_WEIGHTS, a, c(pyr.edge_weights[a], np.float32))
            self.set_field(capi.FIELD_FACE_WEIGHTS, a, c(pyr.face_weights[a], np.float32))

    def set_scene_fields(self, scene):
        """viscosity / density / velocity / solid velocity of a scenes.Scene."""
        def put(kind, axis, v):
            if isinstance(v, (int, float)):
                self.set_field(kind, axis, None, float(v))
            else:
                self.set_field(kind, axis, v.contiguous() if hasattr(v, "contiguous") else np.ascontiguousarray(v))
        put(capi.FIELD_VISCOSITY, 0, scene.viscosity)
        put(capi.FIELD_DENSITY, 0, scene.density)
        for a in range(3):
            put(capi.FIELD_VELOCITY, a, scene.velocity[a])
            if scene.solid_velocity is None:
                self.set_field(capi.FIELD_SOLID_VELOCITY, a, None, 0.0)
            else:
                put(capi.FIELD_SOLID_VELOCITY, a, scene.solid_velocity[a])

    def set_solver_option(self, option, value):
        """avs_set_solver_option: e.g. (capi.OPTION_PRECONDITIONER, capi.PRECONDITIONER_NONE) = plain CG (cpp:638-642)."""
        capi.check(self.lib.avs_set_solver_option(self.h, int(option), int(value)))
        if int(option) == capi.OPTION_F32_VECTORS:
            self.f32_vectors_off = int(value) == 0
            self.f32_vectors_set = int(value) >= 0     # an explicit option overrides AVS_F32_VECTORS in the environment

    # ---- hot path -------------------------------------------------------------------------
    def build_stencils(self):
        capi.check(self.lib.avs_build_stencils(self.h))

    def build_initial_guess(self):
        capi.check(self.lib.avs_build_initial_guess(self.h))

    def build_system(self):
        capi.check(self.lib.avs_build_system(self.h))

    def assemble(self):
        info = capi.AssemblyInfo()
        capi.check(self.lib.avs_assemble(self.h, C.byref(info)))
        self.ainfo = info
        return info

    def solve(self, tol=1e-3, max_iters=2500):
        info = capi.SolveInfo()
        capi.check(self.lib.avs_solve(self.h, float(tol), int(max_iters), C.byref(info)))
        return info

    # ---- multi-GPU ---------------------------------------------------------------------------
    def dist_init(self, rank, world):
        """RCCL communicator: the unique id is made on rank 0 and broadcast with torch.distributed."""
        import torch
        import torch.distributed as dist
        ident = torch.zeros(capi.UNIQUE_ID_BYTES, dtype=torch.uint8)
        if rank == 0:
            buf = (C.c_uint8 * capi.UNIQUE_ID_BYTES)()
            capi.check(self.lib.avs_dist_get_unique_id(buf))
            ident = torch.tensor(list(buf), dtype=torch.uint8)
        dev = torch.device("cuda", torch.cuda.current_device())
        t = ident.to(dev)
        dist.broadcast(t, src=0)
        raw = bytes(t.cpu().tolist())
        buf = (C.c_uint8 * capi.UNIQUE_ID_BYTES).from_buffer_copy(raw)
        capi.check(self.lib.avs_dist_init(self.h, buf, rank, world))

    def dist_init_hosted(self, rank, world):
        """Hosted group (no RCCL communicator): the host program moves the comm-block blobs; here torch.distributed (any backend)
        all-gathers them after every distributed assembly / partition (`_hosted_connect`)."""
        capi.check(self.lib.avs_dist_init_hosted(self.h, rank, world))
        self._hosted = (rank, world)

    def _hosted_connect(self):
        import torch
        import torch.distributed as dist
        rank, world = self._hosted
        blob = (C.c_uint8 * capi.DIST_BLOB_BYTES)()
        capi.check(self.lib.avs_dist_export_blob(self.h, blob))
        mine = torch.tensor(list(bytes(blob)), dtype=torch.uint8)
        if dist.get_backend() == "nccl":
            mine = mine.to(torch.device("cuda", torch.cuda.current_device()))
        allb = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allb, mine)
        raw = b"".join(bytes(t.cpu().tolist()) for t in allb)
        buf = (C.c_uint8 * len(raw)).from_buffer_copy(raw)
        capi.check(self.lib.avs_dist_import_blobs(self.h, buf))   # maps the peers' blocks and runs the transport self-test

    def dist_init_local(self, group, rank):
        capi.check(self.lib.avs_dist_init_local(self.h, group, rank))

    def dist_partition(self, cut_axis=-1):
        capi.check(self.lib.avs_dist_partition(self.h, cut_axis))
        sz = capi.PlanSizes()
        capi.check(self.lib.avs_dist_get_plan_sizes(self.h, C.byref(sz)))
        self.plan_sizes = sz
        self.local_spmv_bytes = 12 * sz.nnz_local + 4 * (sz.n_own + 1) + 16 * sz.n_own
        ti, tb = C.c_int32(), C.c_int32()
        capi.check(self.lib.avs_dist_get_overlap_tiles(self.h, C.byref(ti), C.byref(tb)))
        self.overlap_tiles = (ti.value, tb.value)
        return sz

    def dist_assemble(self, cut_axis=-1):
        """Distributed assembly: this rank assembles only the rows of its slab (instead of assemble + dist_partition)."""
        info = capi.AssemblyInfo()
        capi.check(self.lib.avs_dist_assemble(self.h, cut_axis, C.byref(info)))
        sz = capi.PlanSizes()
        capi.check(self.lib.avs_dist_get_plan_sizes(self.h, C.byref(sz)))
        self.plan_sizes = sz
        self.local_spmv_bytes = 12 * sz.nnz_local + 4 * (sz.n_own + 1) + 16 * sz.n_own
        ti, tb = C.c_int32(), C.c_int32()
        capi.check(self.lib.avs_dist_get_overlap_tiles(self.h, C.byref(ti), C.byref(tb)))
        self.overlap_tiles = (ti.value, tb.value)
        if getattr(self, "_hosted", None) and self._hosted[1] > 1:
            self._hosted_connect()
        return info

    def dist_bind_prepass(self, prepass, cuts, cut_axis=-1):
        """The pre-pass object's next runs are slab-local: this rank's window of the lattices only, global ids through one all-reduce of
        per-tile counts over this context's group (avs_dist_bind_prepass).  cuts: world + 1 fine-cell coordinates, or None (off)."""
        if cuts is None:
            capi.check(self.lib.avs_dist_bind_prepass(self.h, prepass.h, cut_axis, None))
            return
        c = np.ascontiguousarray(cuts, np.int32)
        capi.check(self.lib.avs_dist_bind_prepass(self.h, prepass.h, cut_axis, c.ctypes.data))

    def dist_cuts(self, world, which=0):
        """(cut_axis, cuts[world + 1]) of the last avs_dist_assemble (which = 0) / what its weights suggest for the next frame (1)."""
        cuts = np.empty(world + 1, np.int32)
        ax = C.c_int32()
        capi.check(self.lib.avs_dist_get_cuts(self.h, which, C.byref(ax), cuts.ctypes.data))
        return ax.value, cuts

    def dist_solve(self, tol=1e-3, max_iters=2500):
        info = capi.SolveInfo()
        capi.check(self.lib.avs_dist_solve(self.h, float(tol), int(max_iters), C.byref(info)))
        return info

    def dist_comm_info(self):
        """dict form of avs_dist_info (transport, RCCL rank count, launches / collectives per iteration)."""
        di = capi.DistInfo()
        capi.check(self.lib.avs_dist_get_info(self.h, C.byref(di)))
        return {"world_size": int(di.world_size), "rccl_ranks": int(di.rccl_ranks),
                "transport": {0: "rccl", 1: "direct"}.get(int(di.transport), str(di.transport)),
                "graph_replay": bool(di.graph_replay), "launches_per_iteration": int(di.launches_per_iteration),
                "rccl_calls_per_iteration": int(di.collectives_per_iteration),
                "selftest_rounds": int(di.selftest_rounds), "selftest_bad_entries": int(di.selftest_bad_entries),
                "paranoid": bool(di.paranoid)}

    def dist_solution(self):
        n = self.info().n_velocity
        x = np.empty(n, np.float64)
        capi.check(self.lib.avs_dist_get_solution(self.h, x.ctypes.data, n, capi.MEM_HOST))
        return x

    def dof_table(self, kind=capi.INDEX_VELOCITY):
        i = self.info()
        n = {capi.INDEX_VELOCITY: i.n_velocity, capi.INDEX_EDGE: i.n_edge, capi.INDEX_CENTER: i.n_center}[kind]
        t = np.empty((n, 4), np.int32)
        capi.check(self.lib.avs_get_dof_table(self.h, kind, t.ctypes.data, capi.MEM_HOST))
        return t

    def bench_spmv(self, variant=0, repeats=100):
        """mean time of one launch of the solver's SpMV (ms); checks y against the plain CSR kernel bit for bit.  Probe library only."""
        if not hasattr(self.lib, "avs_bench_spmv") or self.lib is not capi.load(probe=True):
            raise RuntimeError("bench_spmv is a measurement entry of libavs_probe.so: create the solver with probe=True")
        ms = C.c_double()
        capi.check(self.lib.avs_bench_spmv(self.h, variant, repeats, C.byref(ms)))
        return ms.value

    def matrix_format(self):
        """Storage form the solver's SpMV streams (12 / 6 / 4 bytes per non-zero), see avs_matrix_format."""
        fmt = capi.MatrixFormat()
        capi.check(self.lib.avs_get_matrix_format(self.h, C.byref(fmt)))
        return fmt

    def runs_float_vectors(self, info=None):
        """True when a solve of this context iterates on float vectors (avs_pcg_f32.inl): AVS_PRECISION_F32, not switched off by
        AVS_OPTION_F32_VECTORS / AVS_F32_VECTORS=0, and -- given the avs_solve_info of a solve that ran -- not taken by the CU-resident
        loop (which iterates in fp64 on the float system).  Derived from what ran, not from the precision alone."""
        if getattr(self, "precision", 0) != capi.PRECISION_F32 or getattr(self, "f32_vectors_off", False):
            return False
        env = os.environ.get("AVS_F32_VECTORS")
        if env is not None and env.strip() not in ("", "-1") and int(env) == 0 and not getattr(self, "f32_vectors_set", False):
            return False
        return not (info is not None and int(getattr(info, "resident", 0)))

    def spmv_kernel_name(self, info=None):
        fmt = self.matrix_format()
        bpn, tab = int(fmt.bytes_per_nonzero), int(fmt.value_table_size)
        if self.runs_float_vectors(info):   # the float-vector loop (avs_pcg_f32.inl)
            if int(getattr(fmt, "brick_tiles", 0)):
                return (f"k_spmv_brick<DOT,{'VC,' if int(fmt.brick_value_codes) else ''}float> (brick-structured form, float vectors: {int(fmt.brick_tiles)} tiles, "
                        f"{int(fmt.brick_pattern_rows)} rows as {int(fmt.brick_patterns)} geometric row patterns, x of a brick + halo as floats in LDS; {tab}-entry dictionary; brick-major system)")
            return "k_f32_spmv_csr<DOT> (float vectors: column + value code streamed, float products parked in LDS; brick-major system)"
        ltab = "LTAB" if 0 < tab <= 2048 else "GTAB"
        cw = int(fmt.column_windows)
        if int(getattr(fmt, "brick_tiles", 0)):
            return (f"k_spmv_brick<DOT> (brick-structured form: {int(fmt.brick_tiles)} tiles, {int(fmt.brick_pattern_rows)} rows as "
                    f"{int(fmt.brick_patterns)} geometric row patterns (8 B per row), x of a brick + halo in LDS, the other rows as 4-B words; "
                    f"{tab}-entry dictionary; brick-major system)")
        if int(fmt.tile_local_tables):
            form = ("4 B/nnz: tile-local value code | window slot | offset in one word" if cw else
                    "6 B/nnz: 2-B tile-local value codes + int32 column")
            return (f"k_spmv_vi2<512,4096,DOT,WIN=512,TLT=1024{',CWIN' if cw else ''}> ({form}; one value dictionary per 512-row tile "
                    f"staged in LDS, {tab} table entries in total; brick-major system)")
        if cw:
            return (f"k_spmv_vi2<512,4096,DOT,{ltab},WIN=512,CWIN> (4 B/nnz: value code | window slot | offset in one word, "
                    f"{tab}-entry dictionary; brick-major system)")
        return {4: f"k_spmv_vi2<512,4096,DOT,{ltab},PACK,WIN=512> (4 B/nnz packed code|column, brick-major system)",
                6: f"k_spmv_vi2<512,4096,DOT,{ltab},WIN=512> (6 B/nnz value-indexed, brick-major system)",
                12: "k_spmv_tile<512,4096,DOT,VEC,NT> (12 B/nnz, brick-major system)"}.get(bpn, f"{bpn} B/nnz")

    # ---- outputs (numpy, host) ------------------------------------------------------------
    def info(self):
        info = capi.AssemblyInfo()
        capi.check(self.lib.avs_get_assembly_info(self.h, C.byref(info)))
        return info

    def solution(self):
        n = self.info().n_velocity
        x = np.empty(n, np.float64)
        capi.check(self.lib.avs_get_solution(self.h, x.ctypes.data, n, capi.MEM_HOST))
        return x

    def set_solution(self, x):
        """hand the context a solution vector (numpy float64 / torch float64, reference numbering) for the transfer"""
        px, wx = capi.ptr_of(x)
        capi.check(self.lib.avs_set_solution(self.h, px, len(x), wx))

    def initial_guess(self):
        n = self.info().n_velocity
        x = np.empty(n, np.float64)
        capi.check(self.lib.avs_get_initial_guess(self.h, x.ctypes.data, n, capi.MEM_HOST))
        return x

    def csr(self):
        i = self.info()
        rp = np.empty(i.n_velocity + 1, np.int32)
        col = np.empty(i.nnz, np.int32)
        val = np.empty(i.nnz, np.float64)
        rhs = np.empty(i.n_velocity, np.float64)
        capi.check(self.lib.avs_get_csr(self.h, rp.ctypes.data, col.ctypes.data, val.ctypes.data,
                                        rhs.ctypes.data, capi.MEM_HOST))
        return rp, col, val, rhs

    def _stencils(self, fn, ns, nw, cap, bcap):
        cnt = np.empty(ns, np.int32); idx = np.empty((cap, ns), np.int32)
        coef = np.empty((cap, ns), np.float64); bcnt = np.empty(ns, np.int32)
        bval = np.empty((bcap, ns), np.float64); w = np.empty(nw, np.float64)
        capi.check(fn(self.h, cnt.ctypes.data, idx.ctypes.data, coef.ctypes.data, bcnt.ctypes.data,
                      bval.ctypes.data, w.ctypes.data, capi.MEM_HOST))
        return dict(cnt=cnt, idx=idx, coef=coef, bcnt=bcnt, bval=bval, weight=w)

    # ---- post-solve transfer (cpp:655-707) -------------------------------------------------
    def set_regular_index_field(self, axis, idx):
        p, where = capi.ptr_of(idx)
        capi.check(self.lib.avs_set_regular_index_field(self.h, axis, p, where))

    def transfer_to_regular_grid(self):
        """Regular MAC-grid velocity (3 fp32 face grids of the simulation grid, host) after the solve."""
        nx, ny, nz = self.field_res
        outs = [np.empty((nz, ny, nx + 1), np.float32), np.empty((nz, ny + 1, nx), np.float32),
                np.empty((nz + 1, ny, nx), np.float32)]
        capi.check(self.lib.avs_transfer_to_regular_grid(self.h, outs[0].ctypes.data, outs[1].ctypes.data,
                                                         outs[2].ctypes.data, capi.MEM_HOST))
        return outs

    def transfer_to_regular_grid_in_place(self, velocity):
        """In-place form (what the reference does to `vel`, cpp:655-707): `velocity` = three DEVICE tensors that hold the field given to
        set_scalar_field(FIELD_VELOCITY); only the faces the transfer changes are written."""
        ptrs = [capi.ptr_of(v) for v in velocity]
        if any(w != capi.MEM_DEVICE for _, w in ptrs):
            raise ValueError("the in-place transfer updates device arrays")
        capi.check(self.lib.avs_transfer_to_regular_grid_in_place(self.h, ptrs[0][0], ptrs[1][0], ptrs[2][0]))

    def node_grid(self, level):
        nx, ny, nz = (r >> level for r in self.res)
        shp = (nz + 1, ny + 1, nx + 1)
        lab = np.empty(shp, np.int8)
        v = [np.empty(shp, np.float32) for _ in range(3)]
        capi.check(self.lib.avs_get_node_grid(self.h, level, lab.ctypes.data, v[0].ctypes.data, v[1].ctypes.data,
                                              v[2].ctypes.data, capi.MEM_HOST))
        return lab, v

    def edge_stencils(self):
        ne = self.info().n_edge
        return self._stencils(self.lib.avs_get_edge_stencils, ne, ne, capi.EDGE_STENCIL_CAP, capi.EDGE_BOUNDARY_CAP)

    def center_stencils(self):
        nc = self.info().n_center
        return self._stencils(self.lib.avs_get_center_stencils, 3 * nc, nc, capi.CENTER_STENCIL_CAP,
                              capi.CENTER_BOUNDARY_CAP)


def pcg_csr(row_ptr, col, val, b, x0, tol=1e-3, max_iters=2500, device=0):
    """Seam A (avs_pcg_csr) on host numpy arrays."""
    lib = capi.load()
    rp = np.ascontiguousarray(row_ptr, np.int32)
    cl = np.ascontiguousarray(col, np.int32)
    vl = np.ascontiguousarray(val, np.float64)
    bb = np.ascontiguousarray(b, np.float64)
    x = np.array(x0, np.float64, copy=True)
    info = capi.SolveInfo()
    capi.check(lib.avs_pcg_csr(len(bb), rp.ctypes.data, cl.ctypes.data, vl.ctypes.data, bb.ctypes.data,
                               x.ctypes.data, float(tol), int(max_iters), capi.MEM_HOST, device, None,
                               C.byref(info)))
    return x, info


class DevicePrepass:
    """Pre-pass on the GPU (avs_prepass_*): liquid / solid SDF -> weights, label pyramid, index pyramids."""

    def __init__(self, res, dx, levels, n_super=3, extrapolation=0.5, device=0, stream=None, field_res=None):
        """`res`: octree (power-of-two) resolution; `field_res`: the simulation grid the SDFs live on when it is smaller
        (HDK_OctreeGrid::init pads to powers of two, oct.cpp:10-24); None = res."""
        self.lib = capi.load()
        self.res = tuple(int(r) for r in res)
        fr = tuple(int(r) for r in field_res) if field_res is not None else (0, 0, 0)
        self.field_res = tuple(fr[a] or self.res[a] for a in range(3))
        d = capi.PrepassDesc(self.res[0], self.res[1], self.res[2], float(dx), int(levels), int(n_super),
                             float(extrapolation), int(device), C.c_void_p(stream or 0), fr[0], fr[1], fr[2])
        h = C.c_void_p()
        capi.check(self.lib.avs_prepass_create(C.byref(d), C.byref(h)))
        self.h = h
        self.info = None

    def close(self):
        if getattr(self, "h", None):
            self.lib.avs_prepass_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run(self, liquid, solid=None):
        pl, wl = capi.ptr_of(liquid)
        ps, ws = capi.ptr_of(solid)
        if solid is not None and ws != wl:
            raise ValueError("liquid and solid must live in the same memory space")
        capi.check(self.lib.avs_prepass_run(self.h, pl, ps, wl))
        info = capi.PrepassInfo()
        capi.check(self.lib.avs_prepass_get_info(self.h, C.byref(info)))
        self.info = info
        return info

    def set_slab(self, cut_axis, cuts, rank, allreduce):
        """Slab-local mode with the caller's own all-reduce (a hosted group): allreduce(device_pointer, count, stream) sums `count` int32
        in place over all ranks and returns when the result is there.  cuts None: off."""
        if cuts is None:
            self._slab_cb = None
            capi.check(self.lib.avs_prepass_set_slab(self.h, 0, None, 1, 0, capi.ALLREDUCE_I32_FN(0), None))
            return

        def cb(ptr, count, stream, user):
            try:
                allreduce(ptr, count, stream)
                return 0
            except Exception:  # pragma: no cover
                return capi.EINTERNAL if hasattr(capi, "EINTERNAL") else 7
        self._slab_cb = capi.ALLREDUCE_I32_FN(cb)   # (kept alive with the object)
        c = np.ascontiguousarray(cuts, np.int32)
        capi.check(self.lib.avs_prepass_set_slab(self.h, int(cut_axis), c.ctypes.data, len(c) - 1, int(rank), self._slab_cb, None))

    def window(self):
        """(lo[levels], hi[levels], (n_velocity, n_edge, n_center) inside the window) of the last run."""
        lo = np.zeros(8, np.int32)
        hi = np.zeros(8, np.int32)
        nw = np.zeros(3, np.int64)
        capi.check(self.lib.avs_prepass_get_window(self.h, lo.ctypes.data, hi.ctypes.data, nw.ctypes.data))
        L = self.info.levels if self.info is not None else 8
        return lo[:L], hi[:L], tuple(int(v) for v in nw)

    def apply(self, solver):
        capi.check(self.lib.avs_prepass_apply(self.h, solver.h))
        solver.counts = (self.info.n_velocity, self.info.n_edge, self.info.n_center)

    def _shape(self, kind, level, axis):
        r = [self.res[0] >> level, self.res[1] >> level, self.res[2] >> level]
        if kind == 0:
            r[axis] += 1
        elif kind == 1:
            r = [r[a] + (a != axis) for a in range(3)]
        return (r[2], r[1], r[0])

    def labels(self, level):
        out = np.empty(self._shape(2, level, 0), np.int8)
        capi.check(self.lib.avs_prepass_get_labels(self.h, level, out.ctypes.data, capi.MEM_HOST))
        return out

    def mask(self):
        out = np.empty(self._shape(2, 0, 0), np.int8)
        capi.check(self.lib.avs_prepass_get_mask(self.h, out.ctypes.data, capi.MEM_HOST))
        return out

    def index(self, kind, level, axis=0):
        out = np.empty(self._shape(kind, level, axis), np.int32)
        capi.check(self.lib.avs_prepass_get_index(self.h, kind, level, axis, out.ctypes.data, capi.MEM_HOST))
        return out

    def regular_index(self, axis):
        out = np.empty(self._shape(0, 0, axis), np.int32)
        capi.check(self.lib.avs_prepass_get_regular_index(self.h, axis, out.ctypes.data, capi.MEM_HOST))
        return out

    def weights(self, kind, axis=0):
        k = {capi.FIELD_CENTER_WEIGHTS: 2, capi.FIELD_EDGE_WEIGHTS: 1, capi.FIELD_FACE_WEIGHTS: 0}[kind]
        out = np.empty(self._shape(k, 0, axis), np.float32)
        capi.check(self.lib.avs_prepass_get_weights(self.h, kind, axis, out.ctypes.data, capi.MEM_HOST))
        return out
