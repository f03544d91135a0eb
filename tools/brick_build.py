"""Reference builder (torch, CPU or GPU) of the brick-structured SpMV form of csrc/avs_brick.hip -- TEST / MEASUREMENT infrastructure:
the device builder in the library is checked against what this one produces (same SpMV result bit for bit; the arrays may order
patterns differently).  Also: `emulate` = the kernel's arithmetic in torch (left-to-right mul + add per row) for CPU-only tests.

Layout constants mirror avs_internal.hpp (kBrick*)."""
from __future__ import annotations

import numpy as np
import torch

LOFF = [0, 3000, 3648, 3840, 3921]
MAX_ROWS = 1024
MAX_RUNS = 512
BLOCK_WORDS = 1728
XSLOT0 = 3936      # first extra slot (off-lattice columns inside the 27 neighbour bricks)
XSLOTS = 160
RUNLEN = 4   # entries per halo fill run (kBrickRunLen of the library build under test)
PAT_WORDS = 2560
PAT_MAX = 384
PAT_LEN = 64
MIN_ROWS = 64
ETILE_ROWS = 256
STREAMED = 0xFFF


def brick_major(tab, res):
    """dof table (n x 4: level | axis << 8, i, j, k) -> (perm new->old, geometry in the new numbering) as avs_reorder.hip numbers them"""
    nx, ny, nz = res
    tab = tab.long()
    lv = tab[:, 0] & 0xff
    ax = (tab[:, 0] >> 8) & 0xff
    I = [tab[:, 1 + k] for k in range(3)]
    P = [(I[k] << lv) for k in range(3)]
    P = [P[0].clamp(max=nx - 1), P[1].clamp(max=ny - 1), P[2].clamp(max=nz - 1)]
    nbx, nby = (nx + 7) >> 3, (ny + 7) >> 3
    brick = ((P[2] >> 3) * nby + (P[1] >> 3)) * nbx + (P[0] >> 3)
    key = (brick << 9) | ((P[2] & 7) << 6) | ((P[1] & 7) << 3) | (P[0] & 7)
    perm = torch.sort(key, stable=True).indices
    geo = dict(level=lv[perm], axis=ax[perm], i=I[0][perm], j=I[1][perm], k=I[2][perm], brick=brick[perm], nbx=nbx, nby=nby)
    return perm, geo


def permute_csr(rp, col, val, perm):
    """P A P^T with the in-row entry order untouched"""
    dev = rp.device
    n = len(perm)
    inv = torch.empty_like(perm); inv[perm] = torch.arange(n, device=dev)
    lens_old = rp[1:] - rp[:-1]
    lens = lens_old[perm]
    rp_new = torch.zeros(n + 1, dtype=torch.int64, device=dev); rp_new[1:] = torch.cumsum(lens, 0)
    rows_old = torch.repeat_interleave(torch.arange(n, device=dev), lens_old)
    order = torch.sort(inv[rows_old], stable=True).indices
    return rp_new, inv[col[order]], val[order]


def _lat(l, w_tab, off_tab):
    return w_tab[l.clamp(max=4)], off_tab[l.clamp(max=4)]


def build(rp, col, code, geo, table_size, col_bits):
    """rp (n+1, int64), col (nnz, int64), code (nnz, int64): brick-major system; geo: brick_major()[1].  Returns dict of tensors."""
    dev = rp.device
    n = len(rp) - 1
    nnz = len(col)
    lens = rp[1:] - rp[:-1]
    wt = torch.tensor([8, 4, 2, 1, 0], device=dev)
    ot = torch.tensor(LOFF, device=dev)
    lv, ax = geo["level"], geo["axis"]
    I = [geo["i"], geo["j"], geo["k"]]
    brick = geo["brick"]
    nbx, nby = geo["nbx"], geo["nby"]

    # ---- tiles: a brick with >= MIN_ROWS rows is a G tile (cut every MAX_ROWS rows); the rows of runs of smaller bricks are cut into
    #      E tiles of ETILE_ROWS rows
    first = torch.ones(n, dtype=torch.bool, device=dev); first[1:] = brick[1:] != brick[:-1]
    bstart = torch.nonzero(first).flatten()
    brows = torch.diff(torch.cat([bstart, torch.tensor([n], device=dev)]))
    big = brows >= MIN_ROWS
    assert int(brows.max()) < 2048, "a brick with more rows than a run offset can name"
    brick_of_row = torch.cumsum(first.long(), 0) - 1
    rows = torch.arange(n, device=dev)
    big_r = big[brick_of_row]
    in_brick = rows - bstart[brick_of_row]
    # position inside the maximal run of small-brick rows
    small_first = (~big_r) & torch.cat([torch.tensor([True], device=dev), big_r[:-1]])
    sf = torch.nonzero(small_first).flatten()
    run_no = torch.cumsum(small_first.long(), 0) - 1
    in_run = rows - (sf[run_no.clamp(min=0)] if len(sf) else torch.zeros_like(rows))
    tile_first = torch.where(big_r, in_brick % MAX_ROWS == 0, in_run % ETILE_ROWS == 0)
    tile_row0 = torch.nonzero(tile_first).flatten()
    ntiles = len(tile_row0)
    tile_rows = torch.diff(torch.cat([tile_row0, torch.tensor([n], device=dev)]))
    tile_of_row = torch.cumsum(tile_first.long(), 0) - 1
    tile_is_g = big_r[tile_row0]
    tb = brick[tile_row0]  # lattice origin: the tile's brick
    tbx, tby, tbz = tb % nbx, (tb // nbx) % nby, tb // (nbx * nby)

    # ---- per entry: slot of the column on the row's tile lattices
    row_e = torch.repeat_interleave(torch.arange(n, device=dev), lens)
    t_e = tile_of_row[row_e]
    ob = [tbx[t_e], tby[t_e], tbz[t_e]]

    def lattice_slot(l, axis, ijk, ob):
        w, off = _lat(l, wt, ot)
        S = w + 2
        r = [ijk[k] - w * ob[k] + 1 for k in range(3)]
        ok = (l < 4)
        for k in range(3):
            ok = ok & (r[k] >= 0) & (r[k] < S)
        return torch.where(ok, off + ((r[2] * S + r[1]) * S + r[0]) * 3 + axis, torch.full_like(l, -1))

    def base_slot(lr, ijk, ob, lc):
        wl = wt[lr.clamp(max=4)]
        up = (lc - lr).clamp(min=0); dn = (lr - lc).clamp(min=0)
        c = [(((ijk[k] - wl * ob[k]) >> up) << dn) + 1 for k in range(3)]
        wc, off = _lat(lc, wt, ot)
        S = wc + 2
        return off + ((c[2] * S + c[1]) * S + c[0]) * 3

    lc = lv[col]
    cslot = lattice_slot(lc, ax[col], [I[k][col] for k in range(3)], ob)
    base = base_slot(lv[row_e], [I[k][row_e] for k in range(3)], ob, lc)
    # off-lattice columns of G tiles that lie in one of the 27 neighbour bricks (a coarse face on the brick's boundary reads fine faces two
    # cells away): up to XSLOTS extra slots per tile behind the lattice, addressed relative to the row's level-0 base
    cbk = brick[col]
    nbr = ((cbk % nbx - ob[0]).abs() <= 1) & (((cbk // nbx) % nby - ob[1]).abs() <= 1) & ((cbk // (nbx * nby) - ob[2]).abs() <= 1)
    xcand = (cslot < 0) & nbr & tile_is_g[t_e]
    xpair = torch.unique(t_e[xcand] * n + col[xcand])
    xt = xpair // n
    xrank = torch.arange(len(xpair), device=dev) - torch.searchsorted(xt, xt)
    xs_slot = torch.full((nnz,), -1, dtype=torch.int64, device=dev)
    pos = torch.searchsorted(xpair, t_e[xcand] * n + col[xcand])
    xs_slot[xcand] = torch.where(xrank[pos] < XSLOTS, XSLOT0 + xrank[pos], torch.full_like(pos, -1))
    is_x = xs_slot >= 0
    cslot = torch.where(is_x, xs_slot, cslot)
    lc_eff = torch.where(is_x, torch.zeros_like(lc), lc)
    base = torch.where(is_x, base_slot(lv[row_e], [I[k][row_e] for k in range(3)], ob, torch.zeros_like(lc)), base)
    delta = cslot - base
    ent_ok = (cslot >= 0) & (delta >= -4096) & (delta < 4096) & (code < 2048)
    own = lattice_slot(lv, ax, I, [tbx[tile_of_row], tby[tile_of_row], tbz[tile_of_row]])
    bad_rows = torch.zeros(n, dtype=torch.bool, device=dev)
    bad_rows[row_e[~ent_ok]] = True
    # local cell of the row must fit 4 bits per axis after + 1
    wl = wt[lv.clamp(max=4)]
    cell = [I[k] - wl * [tbx, tby, tbz][k][tile_of_row] + 1 for k in range(3)]
    cell_ok = (cell[0] >= 0) & (cell[0] < 16) & (cell[1] >= 0) & (cell[1] < 16) & (cell[2] >= 0) & (cell[2] < 16)
    regular = tile_is_g[tile_of_row] & ~bad_rows & (lens <= PAT_LEN) & (lens > 0) & (own >= 0) & cell_ok & (lv < 4)

    word = ((delta & 0x1fff) << 19) | (lc_eff.clamp(max=3) << 14) | ((code & 0x7ff) << 3)
    j_in_row = torch.arange(nnz, device=dev) - rp[:-1][row_e]
    M1, M2, M3 = -7046029254386353131, -4417276706812531889, 1609587929392839161
    h = (word * M1) ^ (word >> 15) * M2
    h = (h ^ (h >> 29)) * (2 * j_in_row + 1) * M3
    rowh = torch.zeros(n, dtype=torch.int64, device=dev).index_add_(0, row_e, h)
    rowh = rowh * 31 + lens

    # ---- global patterns of the regular rows
    reg_rows = torch.nonzero(regular).flatten()
    gp, ginv = torch.unique(rowh[reg_rows], return_inverse=True)
    npatg = len(gp)
    rep = torch.full((npatg,), n, dtype=torch.int64, device=dev)
    rep.scatter_reduce_(0, ginv, reg_rows, reduce="amin")       # representative row of every pattern (the first)
    plen = lens[rep]
    nz_tag = torch.zeros(n, dtype=torch.bool, device=dev); nz_tag[row_e[lc_eff != 0]] = True
    psimple = (~nz_tag[rep]).long()
    plen4 = (plen + 3) & ~3
    poff = torch.zeros(npatg + 1, dtype=torch.int64, device=dev); poff[1:] = torch.cumsum(plen4, 0)
    pwords = torch.zeros(max(int(poff[-1]), 4), dtype=torch.int64, device=dev)
    # copy the representative rows' words
    src = torch.repeat_interleave(rp[:-1][rep], plen) + (torch.arange(int(plen.sum()), device=dev) - torch.repeat_interleave(torch.cumsum(plen, 0) - plen, plen))
    dst = torch.repeat_interleave(poff[:-1], plen) + (torch.arange(int(plen.sum()), device=dev) - torch.repeat_interleave(torch.cumsum(plen, 0) - plen, plen))
    pwords[dst] = word[src]
    # padding of the last quad: the pattern's first entry's (delta, level) with the code of 0.0 (= table_size)
    first_w = word[rp[:-1][rep]]
    padw = (first_w & ~(0x7ff << 3)) | (table_size << 3)
    for k in (1, 2, 3):
        m = (plen4 - plen) >= k
        pwords[poff[:-1][m] + plen4[m] - k] = padw[m]
    # hash collisions: every regular row must equal its pattern word for word, else it is streamed
    pat_of_row = torch.full((n,), -1, dtype=torch.int64, device=dev)
    pat_of_row[reg_rows] = ginv
    pe = pat_of_row[row_e]
    chk = pe >= 0
    same = torch.ones(nnz, dtype=torch.bool, device=dev)
    same[chk] = pwords[poff[pe[chk]] + j_in_row[chk]] == word[chk]
    coll = torch.zeros(n, dtype=torch.bool, device=dev); coll[row_e[~same]] = True
    regular = regular & ~coll
    pat_of_row[coll] = -1

    # ---- per tile: the patterns it uses, most frequent first; what does not fit the LDS budget is streamed
    reg_rows = torch.nonzero(regular).flatten()
    tp = tile_of_row[reg_rows] * npatg + pat_of_row[reg_rows]
    utp, utp_inv, utp_cnt = torch.unique(tp, return_inverse=True, return_counts=True)
    ut, up = utp // npatg, utp % npatg
    order = torch.sort(ut * (1 << 20) + ((1 << 20) - 1 - utp_cnt.clamp(max=(1 << 20) - 1)), stable=True).indices
    ut, up, utp_cnt = ut[order], up[order], utp_cnt[order]
    rank_of = torch.empty_like(order); rank_of[order] = torch.arange(len(order), device=dev)
    tfirst = torch.searchsorted(ut, ut)
    lidx = torch.arange(len(ut), device=dev) - tfirst
    l4 = plen4[up]
    csum = torch.cumsum(l4, 0)
    lstart = csum - l4 - (csum - l4)[tfirst]
    keep = (lidx < PAT_MAX) & (lstart + l4 <= PAT_WORDS)
    # rows of dropped tile patterns -> streamed
    row_tp = rank_of[utp_inv]
    row_keep = keep[row_tp]
    dropped = reg_rows[~row_keep]
    regular[dropped] = False
    pat_of_row[dropped] = -1
    row_lidx = torch.full((n,), STREAMED, dtype=torch.int64, device=dev)
    row_lidx[reg_rows[row_keep]] = lidx[row_tp[row_keep]]
    ut, up, lstart, lidx = ut[keep], up[keep], lstart[keep], lidx[keep]   # kept entries stay contiguous per tile (dropped ones are a tail)
    npat_tile = torch.bincount(ut, minlength=ntiles)
    pat0 = torch.cumsum(npat_tile, 0) - npat_tile
    pinfo = lstart | ((plen4[up] // 4) << 16) | (psimple[up] << 31)
    # LDS quad list of every tile: quad q of a kept pattern -> its word offset in the global table
    nq = plen4[up] // 4
    qsum = torch.cumsum(nq, 0)
    rep_idx = torch.repeat_interleave(torch.arange(len(up), device=dev), nq)
    qin = torch.arange(int(qsum[-1]) if len(nq) else 0, device=dev) - (qsum - nq)[rep_idx]
    pquads = poff[up][rep_idx] + 4 * qin                      # ordered by (tile, local start): the LDS image is contiguous
    npq_tile = torch.bincount(ut, weights=nq.double(), minlength=ntiles).long()
    pq0 = torch.cumsum(npq_tile, 0) - npq_tile
    if len(pquads) == 0: pquads = torch.zeros(4, dtype=torch.int64, device=dev)
    if len(pinfo) == 0: pinfo = torch.zeros(4, dtype=torch.int64, device=dev)

    # ---- row descriptors of the pattern rows, in EXECUTION order: per tile sorted by (pattern length, tile-local pattern)
    rdesc_all = (row_lidx << 20) | (lv.clamp(max=3) << 18) | (ax << 16) | (cell[2].clamp(0, 15) << 8) | (cell[1].clamp(0, 15) << 4) | cell[0].clamp(0, 15)
    prow = reg_rows[row_keep]
    okey = (tile_of_row[prow] << 40) | (lens[prow] << 28) | (row_lidx[prow] << 12) | (prow - tile_row0[tile_of_row[prow]])
    prow = prow[torch.sort(okey).indices]
    rdesc = rdesc_all[prow]
    rorder = prow - tile_row0[tile_of_row[prow]]
    nprow_tile = torch.bincount(tile_of_row[prow], minlength=ntiles)
    rd0 = torch.cumsum(nprow_tile, 0) - nprow_tile
    if len(rdesc) == 0:
        rdesc = torch.zeros(4, dtype=torch.int64, device=dev); rorder = torch.zeros(4, dtype=torch.int64, device=dev)

    ownslot = torch.where(tile_is_g[tile_of_row] & (own >= 0), own, torch.full_like(own, 0xffff))

    # ---- fill runs: distinct (tile, slot, column) of the regular rows' entries (+ their own slots)
    e_reg = regular[row_e]
    trip = torch.cat([(t_e[e_reg] << 45) | (cslot[e_reg] << 32) | col[e_reg],
                      (tile_of_row[reg_rows[row_keep]] << 45) | (own[reg_rows[row_keep]] << 32) | reg_rows[row_keep]])
    trip = torch.unique(trip)
    trip = trip[tile_of_row[trip & 0xffffffff] != (trip >> 45)]    # halo only: the tile's own rows are filled row by row (ownslot)
    t3, s3, c3 = trip >> 45, (trip >> 32) & 0x1fff, trip & 0xffffffff
    # neighbour brick of every column relative to the tile's brick
    cb = brick[c3]
    cbx, cby, cbz = cb % nbx, (cb // nbx) % nby, cb // (nbx * nby)
    win = (cbz - tbz[t3] + 1) * 9 + (cby - tby[t3] + 1) * 3 + (cbx - tbx[t3] + 1)
    assert bool(((win >= 0) & (win < 27)).all()), "a lattice column outside the 27 neighbour bricks"
    off3 = c3 - bstart[brick_of_row[c3]]
    assert len(off3) == 0 or int(off3.max()) < 2048
    brk = torch.ones(len(trip), dtype=torch.bool, device=dev)
    brk[1:] = (t3[1:] != t3[:-1]) | (s3[1:] != s3[:-1] + 1) | (c3[1:] != c3[:-1] + 1)
    rstart = torch.nonzero(brk).flatten()
    rid = torch.cumsum(brk.long(), 0) - 1
    pos = torch.arange(len(trip), device=dev) - rstart[rid]
    brk = brk | (pos % RUNLEN == 0)
    rstart = torch.nonzero(brk).flatten()
    rlen = torch.diff(torch.cat([rstart, torch.tensor([len(trip)], device=dev)]))
    # a run is 8 B (round 5): the absolute first column | slot << 4 | length - 1
    runs = torch.stack([c3[rstart], (s3[rstart] << 4) | (rlen - 1)], 1).flatten()
    nruns_tile = torch.bincount(t3[rstart], minlength=ntiles)
    run0 = torch.cumsum(nruns_tile, 0) - nruns_tile
    if len(runs) == 0: runs = torch.zeros(4, dtype=torch.int64, device=dev)
    # a slot must never be filled from two different columns
    assert bool(((s3[1:] != s3[:-1]) | (t3[1:] != t3[:-1])).all()), "two columns on one slot"

    # ---- streamed rows: their packed words in CSR order, tile after tile
    srows = torch.nonzero(~regular).flatten()
    st = tile_of_row[srows]
    ns_tile = torch.bincount(st, minlength=ntiles)
    srow0 = torch.cumsum(ns_tile, 0) - ns_tile
    e_s = ~regular[row_e]
    swords = torch.cat([(code[e_s] << col_bits) | col[e_s], torch.zeros(4, dtype=torch.int64, device=dev)])
    slen = lens[srows]
    sstart_g = torch.cumsum(slen, 0) - slen                       # first word of every streamed row (global)
    nsw_tile = torch.bincount(st, weights=slen.double(), minlength=ntiles).long()
    sword0 = torch.zeros(ntiles + 1, dtype=torch.int64, device=dev); sword0[1:] = torch.cumsum(nsw_tile, 0)
    sdesc = torch.stack([(srows - tile_row0[st]) | (slen << 16), sstart_g - sword0[st]], 1)
    if len(sdesc) == 0: sdesc = torch.zeros((1, 2), dtype=torch.int64, device=dev)
    total_words = int(sword0[-1])

    # ---- one descriptor block per tile: header (48 words) | runs | pquads | pinfo | rdesc | rorder (u16) | ownslot (u16), padded to 16 B
    bfirst = torch.full((int(brick.max()) + 2,), 0, dtype=torch.int64, device=dev)   # first row of every brick id (0 for bricks without rows)
    bfirst[brick[bstart]] = bstart
    nb = torch.zeros((ntiles, 32), dtype=torch.int64, device=dev)
    nbz_tot = (int(brick.max()) // (nbx * nby)) + 1
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                X, Y, Z = tbx + dx, tby + dy, tbz + dz
                ok = (X >= 0) & (X < nbx) & (Y >= 0) & (Y < nby) & (Z >= 0) & (Z < nbz_tot)
                idb = ((Z * nby + Y) * nbx + X).clamp(0, len(bfirst) - 1)
                nb[:, (dz + 1) * 9 + (dy + 1) * 3 + (dx + 1)] = torch.where(ok, bfirst[idb], torch.zeros_like(idb))
    z = torch.zeros_like(pat0)
    hdr = torch.cat([torch.stack([tile_row0, tile_rows, npat_tile, nruns_tile, npq_tile, nprow_tile, srow0, ns_tile, sword0[:-1], nsw_tile, rd0, z, z, z, z, z], 1), nb], 1)
    words = 48 + 2 * nruns_tile + npq_tile + npat_tile
    units = (words + 3) // 4
    assert int(words.max()) <= BLOCK_WORDS, f"a tile's descriptor block has {int(words.max())} words"
    assert int(nruns_tile.max()) <= MAX_RUNS, f"a tile with {int(nruns_tile.max())} halo runs"
    unit0 = torch.cumsum(units, 0) - units
    blocks = torch.zeros(int(units.sum()) * 4 + 2048, dtype=torch.int64, device=dev)
    w0 = unit0 * 4
    def put(per_tile_start, counts, values):
        """values: concatenation over tiles (tile-major) of `counts[t]` words each -> blocks[w0[t] + per_tile_start[t] + i]"""
        if len(values) == 0: return
        tix = torch.repeat_interleave(torch.arange(ntiles, device=dev), counts)
        i = torch.arange(len(values), device=dev) - (torch.cumsum(counts, 0) - counts)[tix]
        blocks[w0[tix] + per_tile_start[tix] + i] = values
    blocks[(w0[:, None] + torch.arange(48, device=dev)[None, :]).flatten()] = hdr.flatten()
    o = torch.full_like(w0, 48)
    put(o, 2 * nruns_tile, runs[: 2 * int(nruns_tile.sum())]); o = o + 2 * nruns_tile
    put(o, npq_tile, pquads[: int(npq_tile.sum())]); o = o + npq_tile
    put(o, npat_tile, pinfo[: int(npat_tile.sum())]); o = o + npat_tile
    tile_blk = torch.stack([unit0, units], 1)

    stats = dict(rows=n, nnz=nnz, tiles=ntiles, g_tiles=int(tile_is_g.sum()), regular_rows=int(regular.sum()), regular_nnz=int(lens[regular].sum()),
                 global_patterns=npatg, pattern_words=int(poff[-1]), tile_patterns=int(len(up)), runs=int(len(rstart)), max_runs_per_tile=int(nruns_tile.max()), max_pattern_quads_per_tile=int(npq_tile.max()), streamed_rows=int(len(srows)),
                 streamed_words=total_words, hash_collision_rows=int(coll.sum()), dropped_rows=int(len(dropped)),
                 bytes=dict(rdesc=8 * int(len(prow)), ownslot=2 * n,  swords=4 * total_words, sdesc=8 * int(len(srows)),
                            blocks=16 * int(units.sum()), pwords=4 * int(poff[-1])))
    stats["bytes"]["total"] = sum(stats["bytes"].values())
    i32 = lambda t: t.to(torch.int32).contiguous()
    out = dict(ntiles=ntiles, tile_blk=_u32(tile_blk), blocks=_u32(blocks), rdesc=_u32(torch.stack([rdesc, rorder], 1)),
               ownslot=torch.where(ownslot >= 32768, ownslot - 65536, ownslot).to(torch.int16).contiguous(), pwords=_u32(pwords), sdesc=_u32(sdesc), swords=_u32(swords),
               table_size=table_size, col_bits=col_bits, stats=stats)
    return out


def _u32(t):
    """int64 values in [0, 2^32) -> int32 storage with the same bits"""
    t = t & 0xffffffff
    return torch.where(t >= (1 << 31), t - (1 << 32), t).to(torch.int32).contiguous()


def emulate(form, table, x):
    """y = A x with the arithmetic of k_spmv_brick (torch, any device): per row left-to-right, multiply then add"""
    dev = x.device
    u = lambda t: t.long() & 0xffffffff
    n = len(x)
    y = torch.zeros(n, dtype=torch.float64, device=dev)
    TB = u(form["tile_blk"]); BL = u(form["blocks"]); pw = u(form["pwords"])
    sdesc = u(form["sdesc"]); sw = u(form["swords"])
    cb = form["col_bits"]
    for t in range(form["ntiles"]):
        b0 = int(TB[t, 0]) * 4
        bw = BL[b0: b0 + int(TB[t, 1]) * 4]
        row0, nrows, npat, nruns, npq, nprow, sr0, nsr, sw0, nsw, rd0_ = [int(v) for v in bw[:11]]
        o_runs = 48; o_pq = o_runs + 2 * nruns; o_pi = o_pq + npq
        RD = u(form["rdesc"])
        xs = torch.full((3936 + 160,), float("nan"), dtype=torch.float64, device=dev)
        for q in range(nruns):
            c, d = int(bw[o_runs + 2 * q]), int(bw[o_runs + 2 * q + 1])
            s_, ln = (d >> 4) & 0xfff, (d & 15) + 1
            xs[s_:s_ + ln] = x[c:c + ln]
        if npat > 0:
            for r in range(nrows):
                o = int(form["ownslot"][row0 + r]) & 0xffff
                if o != 0xffff: xs[o] = x[row0 + r]
        lds = torch.zeros(max(4 * npq, 4), dtype=torch.int64, device=dev)
        for q in range(npq):
            o = int(bw[o_pq + q])
            lds[4 * q:4 * q + 4] = pw[o:o + 4]
        for i in range(nprow):
            d = int(RD[rd0_ + i, 0])
            r = int(RD[rd0_ + i, 1])
            pid = d >> 20
            assert pid < npat
            lr, axis = (d >> 18) & 3, (d >> 16) & 3
            c = [(d & 15) - 1, ((d >> 4) & 15) - 1, ((d >> 8) & 15) - 1]
            base = []
            for lc in range(4):
                up, dn = max(lc - lr, 0), max(lr - lc, 0)
                S = (8 >> lc) + 2
                b = [((v >> up) << dn) + 1 for v in c]
                base.append(LOFF[lc] + ((b[2] * S + b[1]) * S + b[0]) * 3)
            pi = int(bw[o_pi + pid])
            off, ln = pi & 0xffff, 4 * ((pi >> 16) & 0x7fff)
            s_ = torch.zeros((), dtype=torch.float64, device=dev)
            for j in range(ln):
                w = int(lds[off + j])
                dl = w >> 19
                if dl >= 4096: dl -= 8192
                code = (w >> 3) & 0x7ff
                tv = table[code] if code < len(table) else torch.zeros((), dtype=torch.float64, device=dev)
                s_ = s_ + tv * xs[base[(w >> 14) & 3] + dl]
            y[row0 + r] = s_
        for i in range(nsr):
            d, stw = int(sdesc[sr0 + i, 0]), int(sdesc[sr0 + i, 1])
            lrow, ln = d & 0xffff, d >> 16
            s_ = torch.zeros((), dtype=torch.float64, device=dev)
            for j in range(ln):
                w = int(sw[sw0 + stw + j])
                s_ = s_ + table[w >> cb] * x[w & ((1 << cb) - 1)]
            y[row0 + lrow] = s_
    return y
