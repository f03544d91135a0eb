"""Host-side partition planner (avs_plan_*, pure integer work): runs without a GPU."""
import numpy as np
import pytest

from adaptiveviscositysolver_amd import capi, scenes
from oracle import oracle as O
from util import oracle_for_scene


@pytest.fixture(scope="module")
def system(built_lib):
    sc = scenes.fat_beam(32, 3)
    o = oracle_for_scene(sc)
    o.prepass()
    o.hot_path()
    A = o.csr()
    return sc, o, A, o.dof_table(O.I_VELOCITY)


@pytest.mark.parametrize("world", [1, 2, 3, 4, 8])
def test_plan_is_consistent(system, world):
    sc, o, A, tab = system
    rp = A.row_ptr.astype(np.int32)
    owner = capi.plan_owners(tab, rp, o.levels, 0, sc.res[0], world)
    assert owner.min() == 0 and owner.max() == world - 1
    # slabs: owner is monotone in the face position along the cut axis, cuts on multiples of 2^(L-1)
    pos = np.minimum(tab[:, 1].astype(np.int64) << (tab[:, 0] & 0xff), sc.res[0] - 1)
    gran = 1 << (o.levels - 1)
    for r in range(world - 1):
        assert (pos[owner == r] // gran).max() < (pos[owner == r + 1] // gran).min()
    plans = [capi.plan_create(rp, A.col, owner, r, world) for r in range(world)]
    # every DOF owned exactly once
    allown = np.concatenate([p["own_global"] for p in plans])
    assert np.array_equal(np.sort(allown), np.arange(A.n))
    # balance (by nnz) within a factor 2 of ideal for these small cases
    nnzs = np.array([len(p["col_local"]) for p in plans])
    assert nnzs.sum() == len(A.col) and nnzs.max() <= 2.0 * nnzs.sum() / world + 4096
    x = np.random.default_rng(1).standard_normal(A.n)
    y_ref = O.spmv_csr(A.row_ptr, A.col, A.val, x)
    for r, p in enumerate(plans):
        n_own = len(p["own_global"])
        # halo grouped by owner ascending, ascending id inside a group; never owned by r
        ho = owner[p["halo_global"]]
        assert (ho != r).all() and (np.diff(ho) >= 0).all()
        for q in np.unique(ho):
            assert (np.diff(p["halo_global"][ho == q]) > 0).all()
        # what I receive from q is exactly what q sends to me, in the same order
        roff = 0
        for i, q in enumerate(p["peers"]):
            cnt = p["recv_counts"][i]
            mine = p["halo_global"][roff:roff + cnt]
            roff += cnt
            pq = plans[q]
            j = list(pq["peers"]).index(r)
            soff = int(pq["send_counts"][:j].sum())
            theirs = pq["own_global"][pq["send_idx"][soff:soff + pq["send_counts"][j]]]
            assert np.array_equal(mine, theirs)
        # local SpMV on [owned | halo] reproduces the owned rows of the global product bit for bit
        x_ext = np.concatenate([x[p["own_global"]], x[p["halo_global"]]])
        y_loc = O.spmv_csr(p["row_ptr_local"].astype(np.int64), p["col_local"], A.val[p["val_src"]], x_ext)
        assert np.array_equal(y_loc, y_ref[p["own_global"]])
        assert p["col_local"].max(initial=-1) < n_own + len(p["halo_global"])


def test_plan_rejects_bad_arguments(built_lib):
    rp = np.array([0, 1], np.int32)
    col = np.array([0], np.int32)
    with pytest.raises(capi.AvsError):
        capi.plan_create(rp, col, np.array([5], np.int32), 0, 2)
    with pytest.raises(capi.AvsError):
        capi.plan_create(rp, col, np.array([0], np.int32), 3, 2)
