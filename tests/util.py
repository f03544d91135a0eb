"""Shared helpers for tests / smoke / bench cpu_baseline: scene -> oracle, comparisons."""
from __future__ import annotations

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402


def _np(t):
    return t.detach().cpu().numpy() if hasattr(t, "detach") else np.asarray(t)


def oracle_for_scene(sc, enhanced=None, f32=False):
    """CPU oracle loaded with the fields of a scenes.Scene (host tensors).  f32: SolveType = fpreal32 (USESINGLEPRECISION)."""
    o = O.Oracle(*sc.res, sc.dx, sc.dt, sc.levels,
                 sc.use_enhanced_gradients if enhanced is None else enhanced, f32=f32)
    o.set_field(O.F_LIQUID, _np(sc.liquid))
    if sc.solid is not None:
        o.set_field(O.F_SOLID, _np(sc.solid))
    for kind, v in ((O.F_VISCOSITY, sc.viscosity), (O.F_DENSITY, sc.density)):
        if isinstance(v, (int, float)):
            o.set_field(kind, None, float(v))
        else:
            o.set_field(kind, _np(v))
    for a in range(3):
        o.set_field(O.F_VELOCITY + a, _np(sc.velocity[a]))
        if sc.solid_velocity is not None:
            o.set_field(O.F_SOLIDVEL + a, _np(sc.solid_velocity[a]))
    return o


def oracle_from_pyramid(sc, pyr):
    """Oracle whose hot-path INPUTS are taken from a prepass.Pyramid (instead of its own pre-pass)."""
    o = oracle_for_scene(sc)
    o.set_levels(pyr.levels)
    o.set_field(O.F_CENTERW, _np(pyr.center_weights))
    for a in range(3):
        o.set_field(O.F_EDGEW + a, _np(pyr.edge_weights[a]))
        o.set_field(O.F_FACEW + a, _np(pyr.face_weights[a]))
    for l in range(pyr.levels):
        o.set_labels(l, _np(pyr.labels[l]))
        for a in range(3):
            o.set_index(O.I_VELOCITY, l, a, _np(pyr.vidx[l][a]))
            o.set_index(O.I_EDGE, l, a, _np(pyr.eidx[l][a]))
        o.set_index(O.I_CENTER, l, 0, _np(pyr.cidx[l]))
    o.finalize_indices()
    return o


def rel_l2(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(np.asarray(b)), 1e-300))


class DevicePyramid:
    """Hot-path inputs produced by the PRODUCT pre-pass (avs_prepass.hip through the C ABI).  Array attributes
    mirror prepass_torch.Pyramid (numpy, downloaded on first use) so that oracle_from_pyramid accepts both."""

    def __init__(self, sc, device=0):
        from adaptiveviscositysolver_amd import DevicePrepass, capi
        self._capi = capi
        # a scene on a non-power-of-two simulation grid: the pre-pass gets field_n* and the SDFs on THAT grid, as from Houdini
        fres = getattr(sc, "field_res", None)
        self.pp = DevicePrepass(sc.res, sc.dx, sc.levels, device=device, field_res=fres)
        if fres is not None and tuple(sc.liquid.shape) == (sc.res[2], sc.res[1], sc.res[0]):
            from adaptiveviscositysolver_amd import scenes as _scenes
            sc = _scenes.crop_to_field(sc)
        info = self.pp.run(sc.liquid, sc.solid)
        self.info = info
        self.levels = int(info.levels)
        self.n_velocity, self.n_edge, self.n_center = int(info.n_velocity), int(info.n_edge), int(info.n_center)
        self._cache = {}

    def _get(self, key, fn):
        if key not in self._cache:
            self._cache[key] = fn()
        return self._cache[key]

    @property
    def labels(self):
        return self._get("labels", lambda: [self.pp.labels(l) for l in range(self.levels)])

    @property
    def vidx(self):
        return self._get("vidx", lambda: [[self.pp.index(0, l, a) for a in range(3)] for l in range(self.levels)])

    @property
    def eidx(self):
        return self._get("eidx", lambda: [[self.pp.index(1, l, a) for a in range(3)] for l in range(self.levels)])

    @property
    def cidx(self):
        return self._get("cidx", lambda: [self.pp.index(2, l, 0) for l in range(self.levels)])

    @property
    def center_weights(self):
        return self._get("cw", lambda: self.pp.weights(self._capi.FIELD_CENTER_WEIGHTS))

    @property
    def edge_weights(self):
        return self._get("ew", lambda: [self.pp.weights(self._capi.FIELD_EDGE_WEIGHTS, a) for a in range(3)])

    @property
    def face_weights(self):
        return self._get("fw", lambda: [self.pp.weights(self._capi.FIELD_FACE_WEIGHTS, a) for a in range(3)])

    def apply(self, solver):
        self.pp.apply(solver)


def build_pyramid(sc, device=0):
    """Pre-pass for a GPU test: the HIP pre-pass of the product (not the torch restatement)."""
    return DevicePyramid(sc, device)


def feed(solver, pyr):
    """hand a pyramid (device pre-pass or prepass_torch.Pyramid) to a ViscositySolve"""
    if hasattr(pyr, "apply"):
        pyr.apply(solver)
    else:
        solver.set_pyramid(pyr)
