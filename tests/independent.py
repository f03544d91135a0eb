"""Checks that share NO code with the row-gather walk of the assembly (cpp:2537-2773 / k_rows / the oracle's
assemble_row): they start from the exported stress-stencil lists only.

(i)  scatter form.  The reference defines the system as  A = M_u + sum_s w_s d_s d_s^T  (cpp:424) and builds it by
     GATHERING, per velocity row, every stress that touches the row (cpp:2537-2745).  Here the same matrix is built
     the other way round: every stress scatters w_s d_s d_s^T (and its boundary terms into the rhs).  If the row
     enumeration misses, duplicates or mis-weights a stress, or picks the wrong list for a face, A - S is not diagonal.
(ii) linear shear known answer.  u = (a y, 0, 0) has the constant strain rate eps_xy = a/2: every complete z-edge
     stencil must give d_s . u = a/2, every other complete edge stencil and every centre stencil 0 -- at T-junctions
     too, which pins the transition coefficients of cpp:1789-1907 (spacing gradientDx, 1/4 and 1/16 child weights).
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp


def scatter_system(n, edge, center, n_center):
    """S = sum_s w_s d_s d_s^T (scipy CSR, duplicates summed) and the boundary part of the rhs, from stencil lists.
    A row's multiplier is the coefficient of its FIRST occurrence in the list (applyToMatrix breaks at the first match,
    cpp:2422-2434); `dups` counts list entries whose face already occurred earlier in the same list (with such
    entries the gathered matrix is still reproduced here, but it is no longer exactly w d d^T)."""
    rows, cols, vals = [], [], []
    rhs = np.zeros(n)
    dups = 0
    for st, per_cell in ((edge, False), (center, True)):
        cnt, idx, coef, bcnt, bval = st["cnt"], st["idx"], st["coef"], st["bcnt"], st["bval"]
        ns = len(cnt)
        if ns == 0:
            continue
        w = st["weight"][np.arange(ns) % n_center] if per_cell else st["weight"]
        cap = idx.shape[0]
        used = np.arange(cap)[:, None] < cnt[None, :]
        bsum = np.where(np.arange(bval.shape[0])[:, None] < bcnt[None, :], bval, 0.0).sum(axis=0)
        first = np.ones((cap, ns), bool)
        for a in range(cap):
            for a2 in range(a):
                first[a] &= ~(used[a2] & (idx[a2] == idx[a]))
        first &= used
        dups += int((used & ~first).sum())
        for a in range(cap):
            if not first[a].any():
                continue
            for b in range(cap):
                mm = first[a] & used[b]
                if mm.any():
                    rows.append(idx[a][mm]); cols.append(idx[b][mm]); vals.append(w[mm] * coef[a][mm] * coef[b][mm])
            mm = first[a]
            np.subtract.at(rhs, idx[a][mm], w[mm] * coef[a][mm] * bsum[mm])
    S = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(n, n)).tocsr()
    S.sum_duplicates()
    return S, rhs, dups


def check_scatter_form(row_ptr, col, val, rhs, x0, edge, center, n_center, rtol=1e-12):
    """A (gathered) minus S (scattered) must be a positive diagonal M, and rhs - rhs_S = M x0."""
    n = len(rhs)
    A = sp.csr_matrix((val, col, row_ptr.astype(np.int64)), shape=(n, n))
    S, rhs_b, dups = scatter_system(n, edge, center, n_center)
    D = (A - S).tocoo()
    scale = max(float(np.abs(val).max()), 1e-300)
    off = D.row != D.col
    off_max = float(np.abs(D.data[off]).max()) if off.any() else 0.0
    assert off_max <= rtol * scale, f"gathered and scattered assembly differ off the diagonal: {off_max:.3e} (scale {scale:.3e})"
    # pattern: every scattered entry is present in the gathered matrix and vice versa (exact zeros by cancellation aside)
    mass = np.asarray((A - S).diagonal()).ravel()
    assert (mass > -rtol * scale).all(), "negative mass term"
    resid = rhs - rhs_b - mass * x0
    rscale = max(float(np.abs(rhs).max()), 1e-300)
    assert float(np.abs(resid).max()) <= 1e-10 * rscale, f"rhs is not M x0 - sum w d b: {np.abs(resid).max():.3e}"
    return dict(off_max=off_max, scale=scale, dups=dups, mass_min=float(mass.min()), mass_max=float(mass.max()),
                mass_positive_fraction=float((mass > 0).mean()))


def face_positions(dof_table, dx):
    """physical centre of every velocity face from the dof table rows (level | axis << 8, i, j, k)"""
    lv = dof_table[:, 0] & 0xff
    ax = (dof_table[:, 0] >> 8) & 0xff
    h = dx * (1 << lv).astype(np.float64)
    pos = (dof_table[:, 1:4].astype(np.float64) + 0.5) * h[:, None]
    pos[np.arange(len(ax)), ax] -= 0.5 * h
    return pos, ax, lv


def check_linear_shear(vel_table, edge_table, dx, edge, center, n_center, labels_fn=None, a=3.0):
    """d_s . u for u = (a y, 0, 0); returns the worst deviations of complete stencils, split by uniform / transition"""
    pos, ax, lv = face_positions(vel_table, dx)
    u = np.where(ax == 0, a * pos[:, 1], 0.0)
    out = {}
    for name, st in (("edge", edge), ("center", center)):
        cnt, idx, coef, bcnt = st["cnt"], st["idx"], st["coef"], st["bcnt"]
        cap = idx.shape[0]
        used = np.arange(cap)[:, None] < cnt[None, :]
        du = np.where(used, coef * u[np.where(used, idx, 0)], 0.0).sum(axis=0)
        if name == "edge":
            e_ax = (edge_table[:, 0] >> 8) & 0xff
            e_lv = edge_table[:, 0] & 0xff
            want = np.where(e_ax == 2, a / 2, 0.0)
            h = dx * (1 << e_lv).astype(np.float64)
            # complete interior stencil of a uniform region: 4 entries, all at the edge's own level, no boundary term
            lv_of = lv[np.where(used, idx, 0)]
            same = np.where(used, lv_of == e_lv[None, :], True).all(axis=0)
            uniform = (cnt == 4) & (bcnt == 0) & same
            # "complete": the four face slots all produced entries (sum of coefficients per gradient axis is 0)
            csum = np.where(used, coef, 0.0).sum(axis=0)
            complete = (bcnt == 0) & (np.abs(csum) <= 1e-9 / h) & (cnt >= 4)
            err = np.abs(du - want)
            out["edge_uniform_max"] = float(err[uniform].max()) if uniform.any() else 0.0
            out["edge_uniform_n"] = int(uniform.sum())
            tr = complete & ~uniform
            out["edge_transition_max"] = float(err[tr].max()) if tr.any() else 0.0
            out["edge_transition_n"] = int(tr.sum())
            out["edge_transition_bad"] = int((err[tr] > 1e-9 * a).sum()) if tr.any() else 0
        else:
            csum = np.where(used, coef, 0.0).sum(axis=0)
            complete = (bcnt == 0) & (cnt >= 2) & (np.abs(csum) <= 1e-9 * np.abs(coef).max())
            out["center_max"] = float(np.abs(du[complete]).max()) if complete.any() else 0.0
            out["center_n"] = int(complete.sum())
    return out


def check_sampled_fields(dx, vel_table, edge_table, center_table, n_center, diag_rho1, diag_rho2, diag_lin, edge_unit, center_unit,
                         edge_lin, center_lin, rho_c0, rho_grad, us_c0, us_grad):
    """(iii) WHERE density and solid velocity are sampled (cpp:2759-2766, 1896-1905, 1952-1960) -- branches a constant field never
    exercises -- pinned without the oracle.  Both fields are LINEAR functions of position stored on their own lattices; trilinear
    interpolation reproduces a linear function exactly, so a sampled value reveals the sample POSITION.

    Three assemblies of one scene: density 1 / density 2 / density rho(P) = rho_c0 + rho_grad . P, and solid velocity (1, 1, 1) /
    u_c(P) = us_c0[c] + us_grad[c] . P.  Then
      * V_i = diag(rho = 2) - diag(rho = 1) is the control volume, and (diag(rho(P)) - diag(rho = 1)) / V_i + 1 must be rho at the
        centre of face i (face_positions: geometry only);
      * the boundary terms of the unit run ARE the coefficients, so bval_lin / bval_unit is the sampled solid velocity: for an edge
        stress of axis a at position E it must be u_a (the EDGE-axis component: the reference's quirk) at
        E + sign/2 dx e_g for a gradient axis g != a; for the centre stress list of axis a at cell centre C it must be u_a at
        C + sign/2 dx e_a."""
    out = {}
    pos, ax, lv = face_positions(vel_table, dx)
    V = diag_rho2 - diag_rho1
    has = V > 1e-9 * np.abs(V).max()
    rho = 1.0 + (diag_lin - diag_rho1)[has] / V[has]
    want = rho_c0 + pos[has] @ np.asarray(rho_grad, np.float64)
    out["density_n"] = int(has.sum())
    out["density_levels"] = sorted(set(int(v) for v in lv[has]))
    out["density_max_rel"] = float(np.abs(rho / want - 1.0).max())
    half = 0.5 * dx
    # edge stresses (boundary terms exist at level 0 only: assert(level == 0), cpp:1903)
    e_ax = (edge_table[:, 0] >> 8) & 0xff
    E = edge_table[:, 1:4].astype(np.float64) * dx
    E[np.arange(len(e_ax)), e_ax] += half
    bad = n = 0
    worst = 0.0
    for k in range(edge_unit["bval"].shape[0]):
        m = edge_unit["bcnt"] > k
        if not m.any():
            continue
        assert np.array_equal(edge_unit["bcnt"], edge_lin["bcnt"])
        cu, cl = edge_unit["bval"][k][m], edge_lin["bval"][k][m]
        a = e_ax[m]
        c0 = np.asarray(us_c0, np.float64)[a]
        g = np.asarray(us_grad, np.float64)[a]                       # gradient of the sampled component u_a
        d = (cl / cu - (c0 + (E[m] * g).sum(axis=1))) / (half * np.sign(cu))
        cand = np.stack([g[np.arange(len(a)), (a + 1) % 3], g[np.arange(len(a)), (a + 2) % 3]], axis=1)
        err = np.abs(cand - d[:, None]).min(axis=1)
        bad += int((err > 2e-3).sum())
        worst = max(worst, float(err.max()))
        n += int(m.sum())
    out.update(edge_boundary_n=n, edge_boundary_bad=bad, edge_boundary_worst=worst)
    Cc = (center_table[:, 1:4].astype(np.float64) + 0.5) * dx
    bad = n = 0
    worst = 0.0
    ns = len(center_unit["bcnt"])
    cell = np.arange(ns) % n_center
    list_axis = np.arange(ns) // n_center
    for k in range(center_unit["bval"].shape[0]):
        m = center_unit["bcnt"] > k
        if not m.any():
            continue
        cu, cl = center_unit["bval"][k][m], center_lin["bval"][k][m]
        a = list_axis[m]
        c0 = np.asarray(us_c0, np.float64)[a]
        g = np.asarray(us_grad, np.float64)[a]
        d = (cl / cu - (c0 + (Cc[cell[m]] * g).sum(axis=1))) / (half * np.sign(cu))
        err = np.abs(d - g[np.arange(len(a)), a])
        bad += int((err > 2e-3).sum())
        worst = max(worst, float(err.max()))
        n += int(m.sum())
    out.update(center_boundary_n=n, center_boundary_bad=bad, center_boundary_worst=worst)
    return out


RHO_LIN = (800.0, (200.0, 800.0, -400.0))
US_C0 = (0.3, 0.6, 0.9)
US_GRAD = ((0.5, -1.0, 2.0), (-2.0, 0.5, 1.0), (1.0, 2.0, -0.5))


def sampled_field_runs(name, n, run):
    """The three assemblies check_sampled_fields needs, through `run(scene) -> dict` (oracle on the CPU, HIP path on the GPU)."""
    from adaptiveviscositysolver_amd import scenes

    def mk():
        if name == "sphere_obstacle":
            return scenes.sphere_with_obstacle(n, 4 if n >= 64 else 3, viscosity=1.0)
        return scenes.fat_beam(n, 3, wall=True, viscosity=1.0)
    sc0 = mk()
    one = scenes.constant_velocity(sc0.res, (1.0, 1.0, 1.0))
    lin = [scenes.linear_field(sc0.res, sc0.dx, a, US_C0[a], US_GRAD[a]) for a in range(3)]
    outs = []
    for dens, us in ((1.0, one), (2.0, one), (scenes.linear_field(sc0.res, sc0.dx, None, RHO_LIN[0], RHO_LIN[1]), lin)):
        sc = mk()
        sc.density, sc.solid_velocity = dens, us
        outs.append(run(sc))

    def diag(o):
        rp, col, val = o["csr"]
        return sp.csr_matrix((val, col, rp.astype(np.int64)), shape=(len(rp) - 1, len(rp) - 1)).diagonal()
    a, b, c = outs
    return check_sampled_fields(sc0.dx, a["vel_table"], a["edge_table"], a["center_table"], a["n_center"], diag(a), diag(b), diag(c),
                                a["edge"], a["center"], c["edge"], c["center"], RHO_LIN[0], RHO_LIN[1], US_C0, US_GRAD)
