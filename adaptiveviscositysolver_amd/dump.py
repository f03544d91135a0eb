"""Field dump ("AVSDUMP1"): the inputs of the hot path as raw little-endian arrays, the exchange
format between a Houdini-side exporter and this library.  Reader: examples/
hotpath_from_dump.cpp.  Layout: magic, nx ny nz levels enhanced (int32), dx dt (f64), n_vel n_edge
n_center (int64); per level: labels int8, vidx[3], eidx[3], cidx int32; then the fields
centre weights, edge weights[3], face weights[3], viscosity, density, velocity[3], solid velocity[3],
each as int32 is_const followed by one float or the dense float array."""
from __future__ import annotations

import struct

import numpy as np


def _np(t, dtype):
    a = t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)
    return np.ascontiguousarray(a, dtype=dtype)


def write_dump(path, scene, pyr):
    with open(path, "wb") as f:
        f.write(b"AVSDUMP1")
        f.write(struct.pack("<5i", scene.res[0], scene.res[1], scene.res[2], pyr.levels,
                            int(bool(scene.use_enhanced_gradients))))
        f.write(struct.pack("<2d", scene.dx, scene.dt))
        f.write(struct.pack("<3q", pyr.n_velocity, pyr.n_edge, pyr.n_center))
        for l in range(pyr.levels):
            f.write(_np(pyr.labels[l], np.int8).tobytes())
            for a in range(3):
                f.write(_np(pyr.vidx[l][a], np.int32).tobytes())
            for a in range(3):
                f.write(_np(pyr.eidx[l][a], np.int32).tobytes())
            f.write(_np(pyr.cidx[l], np.int32).tobytes())

        def field(v):
            if v is None:
                f.write(struct.pack("<if", 1, 0.0))
            elif isinstance(v, (int, float)):
                f.write(struct.pack("<if", 1, float(v)))
            else:
                f.write(struct.pack("<i", 0))
                f.write(_np(v, np.float32).tobytes())
        field(pyr.center_weights)
        for a in range(3):
            field(pyr.edge_weights[a])
        for a in range(3):
            field(pyr.face_weights[a])
        field(scene.viscosity)
        field(scene.density)
        for a in range(3):
            field(scene.velocity[a])
        for a in range(3):
            field(None if scene.solid_velocity is None else scene.solid_velocity[a])
