"""Device pre-pass (avs_prepass_*, HIP kernels) against the oracle's C pre-pass: weights, mask, label
pyramid, index pyramids and DOF counts must agree bit for bit; then the pyramid is handed to a solve
context device-to-device and the assembled system must equal the oracle's."""
import numpy as np
import pytest
import torch

from adaptiveviscositysolver_amd import DevicePrepass, ViscositySolve, capi, scenes
from oracle import oracle as O
from util import build_pyramid, feed, oracle_for_scene

pytestmark = pytest.mark.gpu

CASES = {
    "beam32_L3": lambda dev: scenes.fat_beam(32, 3, device=dev),
    "beam64_wall": lambda dev: scenes.fat_beam(64, 3, wall=True, device=dev),
    "sphere64_L4": lambda dev: scenes.sphere(64, 4, device=dev),
    "noncubic": lambda dev: scenes.fat_beam(64, 3, res=(64, 32, 32), device=dev),
    "sheet64": lambda dev: scenes.thin_sheet(64, 3, thickness_cells=12, device=dev),
    "levels_capped": lambda dev: scenes.fat_beam(16, 6, device=dev),
    "tank64_L3": lambda dev: scenes.tank(64, 3, device=dev),     # liquid on the domain border, collision SDF on the walls
}


@pytest.mark.parametrize("name", list(CASES))
def test_device_prepass_matches_oracle(name, built_lib):
    dev = torch.device("cuda:0")
    sc = scenes.to_device(CASES[name]("cpu"), dev)
    pp = DevicePrepass(sc.res, sc.dx, sc.levels)
    info = pp.run(sc.liquid, sc.solid)
    o = oracle_for_scene(CASES[name]("cpu"))
    o.prepass()
    assert info.levels == o.levels
    assert (info.n_velocity, info.n_edge, info.n_center) == (o.count(0), o.count(1), o.count(2))
    assert np.array_equal(pp.weights(capi.FIELD_CENTER_WEIGHTS).ravel(), o.get_field(O.F_CENTERW))
    for a in range(3):
        assert np.array_equal(pp.weights(capi.FIELD_EDGE_WEIGHTS, a).ravel(), o.get_field(O.F_EDGEW + a))
        assert np.array_equal(pp.weights(capi.FIELD_FACE_WEIGHTS, a).ravel(), o.get_field(O.F_FACEW + a))
    assert np.array_equal(pp.mask(), o.mask())
    for l in range(o.levels):
        assert np.array_equal(pp.labels(l), o.labels(l)), l
        for a in range(3):
            assert np.array_equal(pp.index(capi.INDEX_VELOCITY, l, a), o.index(O.I_VELOCITY, l, a)), (l, a)
            assert np.array_equal(pp.index(capi.INDEX_EDGE, l, a), o.index(O.I_EDGE, l, a)), (l, a)
        assert np.array_equal(pp.index(capi.INDEX_CENTER, l), o.index(O.I_CENTER, l)), l
    # hand over device-to-device and assemble
    s = ViscositySolve(sc.res, sc.dx, sc.dt, info.levels, device=0)
    pp.apply(s)
    s.set_scene_fields(sc)
    s.assemble()
    o.hot_path()
    A = o.csr()
    rp, col, val, rhs = s.csr()
    assert np.array_equal(rp, A.row_ptr.astype(np.int32)) and np.array_equal(col, A.col)
    assert np.array_equal(val, A.val) and np.array_equal(rhs, A.rhs)


def test_device_prepass_full_size_agrees_with_tensor_prepass(built_lib):
    """512^3: the HIP pre-pass and the tensor-op restatement (tests/prepass_torch.py, run on the GPU) give identical pyramids."""
    import prepass_torch
    dev = torch.device("cuda:0")
    sc = scenes.fat_beam(512, 4, device=dev)
    pp = DevicePrepass(sc.res, sc.dx, sc.levels)
    info = pp.run(sc.liquid, sc.solid)
    pyr = prepass_torch.build_pyramid(sc)
    assert info.levels == pyr.levels
    assert (info.n_velocity, info.n_edge, info.n_center) == (pyr.n_velocity, pyr.n_edge, pyr.n_center)
    lib = capi.load()
    for l in range(pyr.levels):
        for kind, grids in ((capi.INDEX_VELOCITY, pyr.vidx), (capi.INDEX_EDGE, pyr.eidx)):
            for a in range(3):
                t = torch.empty_like(grids[l][a])
                capi.check(lib.avs_prepass_get_index(pp.h, kind, l, a, t.data_ptr(), capi.MEM_DEVICE))
                assert torch.equal(t, grids[l][a]), (kind, l, a)
        t = torch.empty_like(pyr.cidx[l])
        capi.check(lib.avs_prepass_get_index(pp.h, capi.INDEX_CENTER, l, 0, t.data_ptr(), capi.MEM_DEVICE))
        assert torch.equal(t, pyr.cidx[l])
    print(f"device pre-pass 512^3: weights {info.weights_ms:.1f} ms, octree {info.octree_ms:.1f} ms, "
          f"classify {info.classify_ms:.1f} ms, numbering {info.number_ms:.1f} ms")


def test_no_liquid_gives_zero_levels(built_lib):
    dev = torch.device("cuda:0")
    liquid = torch.full((16, 16, 16), 10.0, dtype=torch.float32, device=dev)
    pp = DevicePrepass((16, 16, 16), 1 / 16, 3)
    info = pp.run(liquid)
    assert info.levels == 0 and info.n_velocity == 0


def _snapshot(pp, levels):
    out = {"cw": pp.weights(capi.FIELD_CENTER_WEIGHTS).copy(), "mask": pp.mask().copy()}
    for a in range(3):
        out[f"ew{a}"] = pp.weights(capi.FIELD_EDGE_WEIGHTS, a).copy()
        out[f"fw{a}"] = pp.weights(capi.FIELD_FACE_WEIGHTS, a).copy()
        out[f"r{a}"] = pp.regular_index(a).copy()
    for l in range(levels):
        out[f"lab{l}"] = pp.labels(l).copy()
        out[f"c{l}"] = pp.index(capi.INDEX_CENTER, l).copy()
        for a in range(3):
            out[f"v{l}{a}"] = pp.index(capi.INDEX_VELOCITY, l, a).copy()
            out[f"e{l}{a}"] = pp.index(capi.INDEX_EDGE, l, a).copy()
    return out


def _same(a, b):
    assert a.keys() == b.keys()
    for k in a:
        assert np.array_equal(a[k], b[k]), k


def test_temporal_reuse_and_lending_are_invisible(built_lib):
    """Round 5: a pre-pass object that runs frame after frame (i) skips what its allocations already hold from their last filling (far
    weight bricks with an unchanged constant, index tiles outside the last occupancy: avs_prepass.hip "temporal reuse") and (ii) LENDS its
    lattices to the solver context instead of copying them, filling its other set of allocations in the next frame.  Neither may be
    visible: a sequence of DIFFERENT scenes through ONE object -- with applies in between, so that the allocations alternate -- must give
    what a fresh object gives for each scene, bit for bit, and a context must keep the system of the frame it was applied in while the
    pre-pass moves on."""
    dev = torch.device("cuda:0")
    frames = [scenes.fat_beam(64, 3), scenes.thin_sheet(64, 3, thickness_cells=12), scenes.sphere(64, 3), scenes.fat_beam(64, 3, wall=True),
              scenes.tank(64, 3), scenes.fat_beam(64, 3), scenes.fat_beam(64, 3)]
    pp = DevicePrepass((64, 64, 64), frames[0].dx, 3)
    held = []          # (context, the CSR it must keep giving)
    for k, sc_h in enumerate(frames):
        sc = scenes.to_device(sc_h, dev)
        info = pp.run(sc.liquid, sc.solid)
        fresh = DevicePrepass(sc.res, sc.dx, 3)
        finfo = fresh.run(sc.liquid, sc.solid)
        assert (info.levels, info.n_velocity, info.n_edge, info.n_center) == (finfo.levels, finfo.n_velocity, finfo.n_edge, finfo.n_center), k
        _same(_snapshot(pp, info.levels), _snapshot(fresh, finfo.levels))
        fresh.close()
        if k != 2:     # (frame 2 is not applied: the next run then refills the SAME allocations)
            s = ViscositySolve(sc.res, sc.dx, sc.dt, info.levels, device=0)
            pp.apply(s)
            s.set_scene_fields(sc)
            s.assemble()
            held.append((s, sc, [x.copy() for x in s.csr()]))
        # every context applied so far still assembles ITS frame (the pre-pass has not written into what it lent)
        for s, ssc, want in held[-3:]:
            s.assemble()
            for got, w in zip(s.csr(), want):
                assert np.array_equal(got, w), k
    # the oracle for the last frame, through the lent lattices
    o = oracle_for_scene(frames[-1])
    o.prepass()
    o.hot_path()
    A = o.csr()
    rp, col, val, rhs = held[-1][2]
    assert np.array_equal(rp, A.row_ptr.astype(np.int32)) and np.array_equal(col, A.col) and np.array_equal(val, A.val) and np.array_equal(rhs, A.rhs)
    pp.close()          # the contexts keep what they were lent
    s, ssc, want = held[-1]
    s.assemble()
    info = s.solve(1e-8, 3000)
    assert info.converged == 1
    for s, _, _ in held:
        s.close()
