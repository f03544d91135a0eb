"""N>1 path on CPU: world_size-2/3 gloo processes run the partitioned Jacobi-PCG (same plan, same
halo / all-reduce structure as avs_dist.hip) with the oracle's SpMV as the local kernel, and must
reproduce the single-rank solve.  Covers the host logic of the multi-GPU path; the device side of
the same logic is covered on one GPU by tests/test_gpu_dist.py (virtual ranks)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _dist_pcg(rank, world, port, tol, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.dirname(here), here):
        if p not in sys.path:
            sys.path.insert(0, p)
    from adaptiveviscositysolver_amd import capi, scenes
    from oracle import oracle as O
    from util import oracle_for_scene
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sc = scenes.fat_beam(32, 3, variable_viscosity=True)
        o = oracle_for_scene(sc)      # replicated assembly, like avs_dist_partition's input
        o.prepass()
        o.hot_path()
        A = o.csr()
        x0 = o.initial_guess()
        rp32 = A.row_ptr.astype(np.int32)
        owner = capi.plan_owners(o.dof_table(O.I_VELOCITY), rp32, o.levels, 0, sc.res[0], world)
        p = capi.plan_create(rp32, A.col, owner, rank, world)
        own, n_own, n_halo = p["own_global"], len(p["own_global"]), len(p["halo_global"])
        rpl = p["row_ptr_local"].astype(np.int64)
        val = A.val[p["val_src"]]
        b = A.rhs[own]
        x = x0[own].copy()

        def halo_exchange(v_own):
            ext = np.concatenate([v_own, np.zeros(n_halo)])
            reqs, soff, roff, bufs = [], 0, 0, []
            for i, q in enumerate(p["peers"]):
                sc_, rc_ = int(p["send_counts"][i]), int(p["recv_counts"][i])
                snd = torch.from_numpy(np.ascontiguousarray(v_own[p["send_idx"][soff:soff + sc_]]))
                rcv = torch.empty(rc_, dtype=torch.float64)
                reqs.append(dist.isend(snd, int(q)))
                reqs.append(dist.irecv(rcv, int(q)))
                bufs.append((roff, rc_, rcv, snd))
                soff += sc_
                roff += rc_
            for r in reqs:
                r.wait()
            for roff, rc_, rcv, _ in bufs:
                ext[n_own + roff:n_own + roff + rc_] = rcv.numpy()
            return ext

        def allsum(*vals):
            t = torch.tensor(vals, dtype=torch.float64)
            dist.all_reduce(t)
            return t.tolist()

        def spmv(v_own):
            return O.spmv_csr(rpl, p["col_local"], val, halo_exchange(v_own))

        diag = np.ones(n_own)
        for i in range(n_own):
            sl = slice(rpl[i], rpl[i + 1])
            hit = p["col_local"][sl] == i
            if hit.any():
                diag[i] = val[sl][hit][-1]
        invd = np.where(diag != 0, 1.0 / diag, 1.0)
        # Eigen's loop (cpp:618-630), scalars all-reduced
        r = b - spmv(x)
        bb, rr = allsum(float(b @ b), float(r @ r))
        thr = max(tol * tol * bb, np.finfo(float).tiny)
        iters = 0
        if rr >= thr:
            pvec = invd * r
            (rho,) = allsum(float(r @ pvec))
            while iters < 5000:
                t = spmv(pvec)
                (pAp,) = allsum(float(pvec @ t))
                alpha = rho / pAp
                x += alpha * pvec
                r -= alpha * t
                z = invd * r
                rr, rz = allsum(float(r @ r), float(r @ z))
                if rr < thr:
                    break
                beta = rz / rho
                rho = rz
                pvec = z + beta * pvec
                iters += 1
        full = np.zeros(A.n)
        full[own] = x
        tf = torch.from_numpy(full)
        dist.all_reduce(tf)
        if rank == 0:
            xo, io = o.solve(tol, 5000)
            np.savez(os.path.join(out_dir, "res.npz"), x=tf.numpy(), xo=xo, iters=iters, iters_o=io.iterations,
                     err=np.sqrt(rr / bb))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_partitioned_pcg_matches_single_rank(world, tmp_path, built_lib):
    port = _free_port()
    tol = 1e-10
    mp.spawn(_dist_pcg, args=(world, port, tol, str(tmp_path)), nprocs=world, join=True)
    res = np.load(tmp_path / "res.npz")
    rel = np.linalg.norm(res["x"] - res["xo"]) / np.linalg.norm(res["xo"])
    assert rel < 1e-8
    assert abs(int(res["iters"]) - int(res["iters_o"])) <= 3
    assert float(res["err"]) <= tol
