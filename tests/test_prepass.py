"""The torch pre-pass (tests/prepass_torch.py, test infrastructure) against the oracle's C pre-pass: two independent
restatements of cpp:712-1715 + HDK_OctreeGrid.cpp:4-920 must agree bit for bit."""
import numpy as np
import pytest

import prepass_torch as prepass
from adaptiveviscositysolver_amd import scenes
from oracle import oracle as O
from util import oracle_for_scene

CASES = {
    "beam32_L3": lambda: scenes.fat_beam(32, 3),
    "beam32_wall": lambda: scenes.fat_beam(32, 3, wall=True),
    "sphere32_L3": lambda: scenes.sphere(32, 3),
    "sphere64_L4": lambda: scenes.sphere(64, 4),
    "noncubic": lambda: scenes.fat_beam(64, 3, res=(64, 32, 32)),
    "sheet64": lambda: scenes.thin_sheet(64, 3, thickness_cells=12),
    "levels_capped": lambda: scenes.fat_beam(16, 6),
}


@pytest.mark.parametrize("name", list(CASES))
def test_prepass_matches_oracle(name):
    sc = CASES[name]()
    o = oracle_for_scene(sc)
    o.prepass()
    p = prepass.build_pyramid(sc)
    assert p.levels == o.levels
    assert (p.n_velocity, p.n_edge, p.n_center) == (o.count(0), o.count(1), o.count(2))
    assert np.array_equal(p.center_weights.numpy().ravel(), o.get_field(O.F_CENTERW))
    for a in range(3):
        assert np.array_equal(p.edge_weights[a].numpy().ravel(), o.get_field(O.F_EDGEW + a))
        assert np.array_equal(p.face_weights[a].numpy().ravel(), o.get_field(O.F_FACEW + a))
    assert np.array_equal(p.mask.numpy(), o.mask())
    for l in range(o.levels):
        assert np.array_equal(p.labels[l].numpy(), o.labels(l))
        for a in range(3):
            assert np.array_equal(p.vidx[l][a].numpy(), o.index(O.I_VELOCITY, l, a))
            assert np.array_equal(p.eidx[l][a].numpy(), o.index(O.I_EDGE, l, a))
        assert np.array_equal(p.cidx[l].numpy(), o.index(O.I_CENTER, l))


def test_numbering_follows_hdk_tile_order():
    """ids grow along x inside a 16^3 tile, then y, then z, then tile x/y/z (cpp:1566-1593)."""
    sc = scenes.fat_beam(32, 1)
    p = prepass.build_pyramid(sc)
    g = p.vidx[0][1].numpy()          # y faces: 32 x 33 x 32
    tab = {}
    for (k, j, i), v in np.ndenumerate(g):
        if v >= 0:
            tab[int(v)] = (i // 16, j // 16, k // 16, k % 16, j % 16, i % 16)
    keys = [tab[v] for v in sorted(tab)]
    order = [(t[2], t[1], t[0], t[3], t[4], t[5]) for t in keys]     # tile z, y, x, then voxel z, y, x
    assert order == sorted(order)
