"""Field dump ("AVSDUMP1" / "AVSDUMP2"): the inputs of the hot path as raw little-endian arrays, the exchange
format between a Houdini-side exporter and this library.  Reader: examples/
hotpath_from_dump.cpp; the HDK shim writes the same bytes (shim/HDK_AdaptiveViscosity_avs.cpp).
Layout: magic, nx ny nz levels enhanced (int32) [AVSDUMP2: + field_nx field_ny field_nz (int32): the SIMULATION grid,
smaller than the power-of-two octree lattice nx ny nz -- HDK_OctreeGrid::init pads, oct.cpp:10-24 -- i.e. every real frame],
dx dt (f64), n_vel n_edge n_center (int64); per level: labels int8, vidx[3], eidx[3], cidx int32 (octree lattices); then
the fields centre weights, edge weights[3], face weights[3], viscosity, density, velocity[3], solid velocity[3], each as
int32 is_const followed by one float or the dense float array -- on the octree lattice in AVSDUMP1, on the SIMULATION grid's
lattices in AVSDUMP2 (what avs_set_scalar_field takes from a context created with field_n*)."""
from __future__ import annotations

import struct

import numpy as np


def _np(t, dtype):
    a = t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)
    return np.ascontiguousarray(a, dtype=dtype)


def write_dump(path, scene, pyr, field_res=None):
    """`field_res` (or scene.field_res): write AVSDUMP2 with the scalar fields cropped to the simulation grid."""
    fres = field_res if field_res is not None else getattr(scene, "field_res", None)
    if fres is not None and tuple(fres) == tuple(scene.res):
        fres = None

    def crop(a, kind, axis):
        if fres is None:
            return a
        r = [int(fres[0]), int(fres[1]), int(fres[2])]
        if kind == 0:
            r[axis] += 1
        elif kind == 1:
            r = [r[b] + (b != axis) for b in range(3)]
        return np.ascontiguousarray(a[:r[2], :r[1], :r[0]])          # (already on the simulation grid: a no-op)
    with open(path, "wb") as f:
        f.write(b"AVSDUMP1" if fres is None else b"AVSDUMP2")
        f.write(struct.pack("<5i", scene.res[0], scene.res[1], scene.res[2], pyr.levels,
                            int(bool(scene.use_enhanced_gradients))))
        if fres is not None:
            f.write(struct.pack("<3i", *[int(v) for v in fres]))
        f.write(struct.pack("<2d", scene.dx, scene.dt))
        f.write(struct.pack("<3q", pyr.n_velocity, pyr.n_edge, pyr.n_center))
        for l in range(pyr.levels):
            f.write(_np(pyr.labels[l], np.int8).tobytes())
            for a in range(3):
                f.write(_np(pyr.vidx[l][a], np.int32).tobytes())
            for a in range(3):
                f.write(_np(pyr.eidx[l][a], np.int32).tobytes())
            f.write(_np(pyr.cidx[l], np.int32).tobytes())

        def field(v, kind=2, axis=0):
            if v is None:
                f.write(struct.pack("<if", 1, 0.0))
            elif isinstance(v, (int, float)):
                f.write(struct.pack("<if", 1, float(v)))
            else:
                f.write(struct.pack("<i", 0))
                f.write(crop(_np(v, np.float32), kind, axis).tobytes())
        field(pyr.center_weights)
        for a in range(3):
            field(pyr.edge_weights[a], 1, a)
        for a in range(3):
            field(pyr.face_weights[a], 0, a)
        field(scene.viscosity)
        field(scene.density)
        for a in range(3):
            field(scene.velocity[a], 0, a)
        for a in range(3):
            field(None if scene.solid_velocity is None else scene.solid_velocity[a], 0, a)
