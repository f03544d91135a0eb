// avs_pcg_f32.inl -- the float-vector PCG loop of AVS_PRECISION_F32 contexts (included by avs_pcg.hip, inside namespace avs).
//
// The reference built with USESINGLEPRECISION (HDK_Utilities.h:25-37: SolveType = fpreal32, Vector = Eigen::VectorXf) hands a
// SparseMatrix<float> to Eigen::ConjugateGradient (HDK_AdaptiveViscosity.cpp:613-630): matrix values, vectors AND scalars are floats.
// Until round 5 this library assembled that float system bit for bit and then iterated on it in fp64; here the iteration itself runs on
// float vectors:
//   * x, r, p, A p and the inverse diagonal are float arrays (4 B per row and stream instead of 8: the vector kernels of the fp64 loop
//     run at the HBM roofline, so this halves their time);
//   * alpha = absNew / p.Ap, beta = absNew / absOld and the threshold tol^2 |b|^2 are computed in float from float operands;
//   * every row sum is a float sum left to right in the stored column order (one multiply, one add per entry, no FMA), exactly
//     the oracle's SPMV_F (oracle/avs_oracle.c: orc_pcg_csr_f32);
//   * dot products: a thread's own terms are added in float, everything across threads, workgroups and launches in double, and the
//     total is rounded to float where Eigen would hold a float.  Eigen's own (vectorised, 4-accumulator) reduction order is not
//     reproduced -- neither is it by the oracle, whose float dots run left to right --, so iteration counts agree with the oracle's
//     to a few per cent and the solution to the accuracy float CG reaches, not bit for bit.
// The SpMV is the brick kernel instantiated for float vectors where the matrix has the form (k_spmv_brick<DOT, VC, float>: the lattice
// in LDS is half as large), else a plain streaming kernel over the CSR arrays (value index or 8-B values).
// Same control flow as the fp64 launch-per-phase loop (pcg_solve): three launches per iteration with the scalar steps fused into the
// vector kernels, chunks of kChunk iterations replayed from a captured hipGraph, avs_cancel polled between chunks.

typedef float f4_t __attribute__((ext_vector_type(4)));
typedef unsigned u2_t __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(kBlock) void k_f32_narrow(int64_t n, const double *__restrict__ src, float *__restrict__ dst)
{
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) dst[i] = (float)src[i];
}
__global__ __launch_bounds__(kBlock) void k_f32_widen(int64_t n, const float *__restrict__ src, double *__restrict__ dst)
{
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) dst[i] = (double)src[i];
}

// DiagonalPreconditioner<float>::factorize: invdiag(j) = A(j,j) != 0 ? 1.f / A(j,j) : 1.f
__global__ __launch_bounds__(kBlock) void k_f32_inv_diag(CsrView A, float *__restrict__ invd)
{
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= A.n) return;
    float d = 0.f;
    for (int k = A.row_ptr[i]; k < A.row_ptr[i + 1]; ++k)
        if (A.col[k] == (int32_t)i) d = (float)(A.val ? A.val[k] : A.table[(A.tab_ptr ? A.tab_ptr[i / kTileRows] : 0) + A.codes[k]]);
    invd[i] = (d != 0.f && !A.no_precond) ? 1.f / d : 1.f;
}
// one small dictionary: invd[i] == invtab[dcode[i]] (dcode: k_inv_diag_coded), the table inverted in float
__global__ __launch_bounds__(kBlock) void k_f32_invtab(CsrView A, float *__restrict__ invtab)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i > A.table_size) return;
    const float d = i < A.table_size ? (float)A.table[i] : 0.f;
    invtab[i] = (d != 0.f && !A.no_precond) ? 1.f / d : 1.f;
}

// Eigen's float threshold: tol^2 |b|^2 in float, at least the smallest normal float (ConjugateGradient.h: considerAsZero =
// (std::numeric_limits<RealScalar>::min)()); b.b and r.r arrive as double sums of float terms
__global__ void k_f32_threshold(PcgScalars *sc, double tol)
{
    if (threadIdx.x != 0 || blockIdx.x != 0 || sc->done == 3) return;
    const float rhs = (float)sc->rhs_norm2, t = (float)tol;
    float thr = t * t * rhs;
    if (thr < 1.17549435e-38f) thr = 1.17549435e-38f;
    sc->threshold = (double)thr;
    sc->rhs_norm2 = (double)rhs;
    sc->rr = (double)(float)sc->rr;
    sc->done = ((float)sc->rr < thr) ? 1 : 0;
}

// y = A x for any CsrView (8-B values, one dictionary, tile-local dictionaries): coalesced stream of (column, value) -> float products
// parked in LDS -> every row adds its segment left to right.  256 rows per workgroup (a workgroup lies inside one 512-row tile: its
// dictionary base is uniform).  The systems that land here are small (no brick form: < 2 M rows) and cache-resident.
template <bool DOT>
__global__ __launch_bounds__(kBlock) void k_f32_spmv_csr(CsrView A, const float *__restrict__ x, float *__restrict__ y,
                                                         double *__restrict__ partial, const PcgScalars *sc)
{
    if (DOT && sc && sc->done) return;
    __shared__ float prod[kStreamCap];
    __shared__ double red[4];
    const int tid = threadIdx.x;
    const int64_t row0 = (int64_t)blockIdx.x * kBlock;
    const int64_t row = row0 + tid;
    const int64_t rlast = (row0 + kBlock < A.n) ? row0 + kBlock : A.n;
    const int s_blk = A.row_ptr[row0];
    const int e_blk = A.row_ptr[rlast];
    const int tbase = A.tab_ptr ? A.tab_ptr[row0 / kTileRows] : 0;
    int rs = 0, re = 0;
    if (row < A.n) {
        rs = A.row_ptr[row];
        re = A.row_ptr[row + 1];
    }
    float sum = 0.f;
    for (int ts = s_blk; ts < e_blk; ts += kStreamCap) {
        const int te = (ts + kStreamCap < e_blk) ? ts + kStreamCap : e_blk;
        for (int k0 = ts + tid; k0 < te; k0 += 4 * kBlock) {
            int c[4];
            float v[4], xv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = k0 + u * kBlock < te ? k0 + u * kBlock : ts;
                c[u] = A.col[k];
                v[u] = A.val && !A.codes ? (float)A.val[k] : (float)A.table[tbase + A.codes[k]];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) xv[u] = x[c[u]];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (k0 + u * kBlock < te) prod[k0 + u * kBlock - ts] = v[u] * xv[u];
        }
        __syncthreads();
        const int a = rs > ts ? rs : ts;
        const int b = re < te ? re : te;
        for (int j = a; j < b; ++j) sum += prod[j - ts];
        __syncthreads();
    }
    if (row < A.n) y[row] = sum;
    if (DOT) {
        double d = (row < A.n) ? (double)(sum * x[row]) : 0.;
        d = block_sum(d, red);
        if (tid == 0) partial[blockIdx.x] = d;
    }
}

// r = b - t ; partials: [0..g) b.b, [g..2g) r.r
__global__ __launch_bounds__(kBlock) void k_f32_init_residual(int64_t n, const float *__restrict__ b, const float *__restrict__ t,
                                                              float *__restrict__ r, double *__restrict__ partial)
{
    __shared__ double red[4];
    float bb = 0.f, rr = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const float bi = b[i];
        const float ri = bi - t[i];
        r[i] = ri;
        bb += bi * bi;
        rr += ri * ri;
    }
    const double sb = block_sum((double)bb, red);
    const double sr = block_sum((double)rr, red);
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = sb;
        partial[gridDim.x + blockIdx.x] = sr;
    }
}

// p = invd * r ; partial r.p
template <bool CODED>
__global__ __launch_bounds__(kBlock) void k_f32_init_p(int64_t n, const float *__restrict__ r, const float *__restrict__ invd,
                                                       const uint16_t *__restrict__ dcode, float *__restrict__ p, float *__restrict__ x,
                                                       double *__restrict__ partial, const PcgScalars *sc)
{
    __shared__ double red[4];
    const int done = sc->done;
    float rz = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        if (done == 3) { x[i] = 0.f; continue; } // rhsNorm2 == 0 -> x.setZero()
        if (done) continue;
        const float ri = r[i];
        const float zi = (CODED ? invd[dcode[i]] : invd[i]) * ri;
        p[i] = zi;
        rz += ri * zi;
    }
    const double s = block_sum((double)rz, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// r -= alpha t ; partials r.r and r.(invd r).  Four rows per thread (16-B accesses).  FUSED: every workgroup folds the SpMV's `nb`
// partial sums itself (the alpha step, as in k_update_r); else OP_ALPHA has left p.Ap in sc->pAp.  Either way alpha is the FLOAT
// quotient of the float-rounded sums.
template <bool CODED, bool FUSED, bool KEEP>
__global__ __launch_bounds__(kBlock) void k_f32_update_r(int64_t n, float *__restrict__ r, const float *__restrict__ t,
                                                         const float *__restrict__ invd, const uint16_t *__restrict__ dcode, PcgScalars *sc,
                                                         double *__restrict__ partial, const double *__restrict__ spmv_partial, int nb, int parity)
{
    if (sc->done) {
        if (FUSED && blockIdx.x == 0 && threadIdx.x == 0 && sc->done == 2) sc->done = 1; // the pending x update has run (OP_ALPHA)
        return;
    }
    __shared__ double red[4];
    float alpha;
    if (FUSED) {
        __shared__ double tot;
        double pap = 0.;
        for (int k = threadIdx.x; k < nb; k += kBlock) pap += spmv_partial[k];
        pap = block_sum(pap, red);
        if (threadIdx.x == 0) tot = pap;
        __syncthreads();
        pap = tot;
        alpha = (float)(parity ? sc->rho_alt : sc->rho) / (float)pap;
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            sc->red[0] = pap;
            sc->pAp = (double)(float)pap;
            sc->alpha = (double)alpha;
        }
    } else {
        alpha = (float)(parity ? sc->rho_alt : sc->rho) / (float)sc->pAp;
        if (blockIdx.x == 0 && threadIdx.x == 0) sc->alpha = (double)alpha; // (OP_ALPHA divided in double: k_f32_update_xp reads this one)
    }
    float rr = 0.f, rz = 0.f;
    const int64_t n4 = n >> 2;
    for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < n4; j += (int64_t)gridDim.x * kBlock) {
        const int64_t i = 4 * j;
        const f4_t rv = *reinterpret_cast<const f4_t *>(r + i);
        const f4_t tv = stream_load_k<KEEP>(reinterpret_cast<const f4_t *>(t + i));
        f4_t iv;
        if (CODED) {
            const u2_t cc = stream_load_k<KEEP>(reinterpret_cast<const u2_t *>(dcode + i));
            iv.x = invd[cc.x & 0xffffu]; iv.y = invd[cc.x >> 16]; iv.z = invd[cc.y & 0xffffu]; iv.w = invd[cc.y >> 16];
        } else iv = *reinterpret_cast<const f4_t *>(invd + i);
        f4_t rn;
        rn.x = rv.x - alpha * tv.x; rn.y = rv.y - alpha * tv.y; rn.z = rv.z - alpha * tv.z; rn.w = rv.w - alpha * tv.w;
        *reinterpret_cast<f4_t *>(r + i) = rn;
        rr += rn.x * rn.x; rz += rn.x * (iv.x * rn.x);
        rr += rn.y * rn.y; rz += rn.y * (iv.y * rn.y);
        rr += rn.z * rn.z; rz += rn.z * (iv.z * rn.z);
        rr += rn.w * rn.w; rz += rn.w * (iv.w * rn.w);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int64_t i = n4 * 4; i < n; ++i) {
            const float ri = r[i] - alpha * t[i];
            r[i] = ri;
            rr += ri * ri;
            rz += ri * ((CODED ? invd[dcode[i]] : invd[i]) * ri);
        }
    const double srr = block_sum((double)rr, red);
    const double srz = block_sum((double)rz, red);
    if (threadIdx.x == 0) {
        partial[blockIdx.x] = srr;
        partial[gridDim.x + blockIdx.x] = srz;
    }
}

// x += alpha p (also in the iteration that converges: done == 2), then p = invd r + beta p.  The beta step -- fold of k_f32_update_r's
// 2 g partials, convergence test, beta = absNew / absOld in float -- is done here by every workgroup for itself (as k_update_xp<FUSED>).
template <bool CODED, bool KEEP>
__global__ __launch_bounds__(kBlock) void k_f32_update_xp(int64_t n, float *__restrict__ x, float *__restrict__ p, const float *__restrict__ r,
                                                          const float *__restrict__ invd, const uint16_t *__restrict__ dcode, PcgScalars *sc,
                                                          const double *__restrict__ partial, int g, int parity)
{
    int done = sc->done;
    if (done == 1 || done == 3) return;
    const float alpha = (float)sc->alpha;
    float beta = 0.f;
    if (done == 0) {
        __shared__ double red[4], tot[2];
        double rr = 0., rz = 0.;
        for (int i = threadIdx.x; i < g; i += kBlock) {
            rr += partial[i];
            rz += partial[g + i];
        }
        rr = block_sum(rr, red);
        rz = block_sum(rz, red);
        if (threadIdx.x == 0) { tot[0] = rr; tot[1] = rz; }
        __syncthreads();
        const float rrf = (float)tot[0], rzf = (float)tot[1];
        const float absOld = (float)(parity ? sc->rho_alt : sc->rho);
        if (rrf < (float)sc->threshold) done = 2; // Eigen: break before i++ (x += alpha p still pending)
        else beta = rzf / absOld;
        if (blockIdx.x == 0 && threadIdx.x == 0) { // what OP_BETA does
            sc->red[0] = (double)rrf;
            sc->red[1] = (double)rzf;
            sc->rr = (double)rrf;
            if (done == 2) sc->done = 2;
            else {
                if (parity) sc->rho = (double)rzf;
                else sc->rho_alt = (double)rzf;
                sc->beta = (double)beta;
                sc->iter += 1;
            }
        }
    }
    if (done == 2) {
        for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) x[i] += alpha * p[i];
        return;
    }
    const int64_t n4 = n >> 2;
    for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < n4; j += (int64_t)gridDim.x * kBlock) {
        const int64_t i = 4 * j;
        const f4_t pv = *reinterpret_cast<const f4_t *>(p + i);
        const f4_t xv = stream_load_k<KEEP>(reinterpret_cast<const f4_t *>(x + i));
        const f4_t rv = stream_load_k<KEEP>(reinterpret_cast<const f4_t *>(r + i));
        f4_t iv;
        if (CODED) {
            const u2_t cc = stream_load_k<KEEP>(reinterpret_cast<const u2_t *>(dcode + i));
            iv.x = invd[cc.x & 0xffffu]; iv.y = invd[cc.x >> 16]; iv.z = invd[cc.y & 0xffffu]; iv.w = invd[cc.y >> 16];
        } else iv = *reinterpret_cast<const f4_t *>(invd + i);
        f4_t xn, pn;
        xn.x = xv.x + alpha * pv.x; xn.y = xv.y + alpha * pv.y; xn.z = xv.z + alpha * pv.z; xn.w = xv.w + alpha * pv.w;
        pn.x = iv.x * rv.x + beta * pv.x; pn.y = iv.y * rv.y + beta * pv.y; pn.z = iv.z * rv.z + beta * pv.z; pn.w = iv.w * rv.w + beta * pv.w;
        stream_store_k<KEEP>(xn, reinterpret_cast<f4_t *>(x + i));
        *reinterpret_cast<f4_t *>(p + i) = pn;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int64_t i = n4 * 4; i < n; ++i) {
            const float pi = p[i];
            x[i] += alpha * pi;
            p[i] = (CODED ? invd[dcode[i]] : invd[i]) * r[i] + beta * pi;
        }
}

template <bool DOT>
static avs_status spmv_f32_dispatch(const CsrView &A, const float *x, float *y, double *partial, const PcgScalars *sc, hipStream_t stream, int *nblocks)
{
    if (A.n <= 0) { if (nblocks) *nblocks = 0; return AVS_OK; }
    if (A.brick && A.brick->ntiles > 0 && A.brick->pwords32) {
        if (nblocks) *nblocks = brick_partial_count(*A.brick, 4);
        return spmv_brick_launch_f32(*A.brick, x, y, DOT ? partial : nullptr, (DOT && sc) ? &sc->done : nullptr, stream);
    }
    const int g = stream_grid(A.n);
    hipLaunchKernelGGL((k_f32_spmv_csr<DOT>), dim3(g), dim3(kBlock), 0, stream, A, x, y, partial, sc);
    if (nblocks) *nblocks = g;
    AVS_HIP(hipGetLastError());
    return AVS_OK;
}

// (KEEP is a template parameter of the vector kernels: see stream_load_k)
#define AVS_F32_LAUNCH_R(C, F, ...)                                                                                   \
    do {                                                                                                              \
        if (keep) hipLaunchKernelGGL((k_f32_update_r<C, F, true>), dim3(g), dim3(kBlock), 0, stream, __VA_ARGS__);   \
        else hipLaunchKernelGGL((k_f32_update_r<C, F, false>), dim3(g), dim3(kBlock), 0, stream, __VA_ARGS__);       \
    } while (0)
#define AVS_F32_LAUNCH_XP(C, ...)                                                                                     \
    do {                                                                                                              \
        if (keep) hipLaunchKernelGGL((k_f32_update_xp<C, true>), dim3(g), dim3(kBlock), 0, stream, __VA_ARGS__);     \
        else hipLaunchKernelGGL((k_f32_update_xp<C, false>), dim3(g), dim3(kBlock), 0, stream, __VA_ARGS__);         \
    } while (0)

// b, x: the context's fp64 arrays (float values); x holds the initial guess and receives the solution (float values again)
static avs_status pcg_solve_f32(PcgWork *w, const CsrView &A, const double *b, double *x, double tol, int max_iters, hipStream_t stream,
                                avs_solve_info *info)
{
    const int64_t n = A.n;
    const size_t na = (size_t)n + 8;
    if (!w->f_x.p) {
        AVS_TRY(w->f_x.alloc(na)); AVS_TRY(w->f_r.alloc(na)); AVS_TRY(w->f_p.alloc(na)); AVS_TRY(w->f_t.alloc(na)); AVS_TRY(w->f_b.alloc(na));
    }
    const bool brick = A.brick && A.brick->ntiles > 0 && A.brick->pwords32;
    {
        size_t need = 2 * ((size_t)((n + kBlock - 1) / kBlock) + 16) + 4 * (size_t)kVecGrid + 16; // the streaming kernel: one partial per 256 rows
        if (brick) { const size_t nb = 2 * ((size_t)A.brick->ntiles * 8 + 16) + 4 * (size_t)kVecGrid + 16; need = nb > need ? nb : need; }
        if (need > w->npartial) {
            AVS_TRY(w->partial.alloc(need));
            w->npartial = need;
            if (w->graph) { (void)hipGraphExecDestroy(w->graph); w->graph = nullptr; }
        }
    }
    const int vgrid = (int)((n + kBlock - 1) / kBlock < kVecGrid ? (n + kBlock - 1) / kBlock : kVecGrid);
    const int g = vgrid > 0 ? vgrid : 1;
    const int rowgrid = (int)((n + kBlock - 1) / kBlock) > 0 ? (int)((n + kBlock - 1) / kBlock) : 1;
    float *xf = w->f_x.p, *p = w->f_p.p, *r = w->f_r.p, *t = w->f_t.p, *bf = w->f_b.p;
    double *partial = w->partial.p;
    PcgScalars *sc = w->sc.p;

    AVS_HIP(hipMemsetAsync(sc, 0, sizeof(PcgScalars), stream));
    hipLaunchKernelGGL(k_f32_narrow, dim3(g), dim3(kBlock), 0, stream, n, b, bf);
    hipLaunchKernelGGL(k_f32_narrow, dim3(g), dim3(kBlock), 0, stream, n, (const double *)x, xf);
    const bool coded = A.codes && !A.tab_ptr && A.table_size <= kViLdsTable;
    float *invd = nullptr;
    if (coded) {
        if (!w->dcode.p) AVS_TRY(w->dcode.alloc((size_t)n + 8));
        if (!w->invtab.p) AVS_TRY(w->invtab.alloc((size_t)kViLdsTable + 1));
        if (!w->f_invtab.p) AVS_TRY(w->f_invtab.alloc((size_t)kViLdsTable + 1));
        const int cg = (int)(((n > A.table_size + 1 ? n : A.table_size + 1) + kBlock - 1) / kBlock);
        hipLaunchKernelGGL(k_inv_diag_coded, dim3(cg), dim3(kBlock), 0, stream, A, w->dcode.p, w->invtab.p);
        hipLaunchKernelGGL(k_f32_invtab, dim3((A.table_size + kBlock) / kBlock), dim3(kBlock), 0, stream, A, w->f_invtab.p);
        invd = w->f_invtab.p;
    } else {
        if (!w->f_invd.p) AVS_TRY(w->f_invd.alloc(na));
        hipLaunchKernelGGL(k_f32_inv_diag, dim3(rowgrid), dim3(kBlock), 0, stream, A, w->f_invd.p);
        invd = w->f_invd.p;
    }
    const uint16_t *dcode = coded ? w->dcode.p : nullptr;
    AVS_HIP(hipEventRecord(w->ev0, stream));

    AVS_TRY(spmv_f32_dispatch<false>(A, xf, t, nullptr, nullptr, stream, nullptr));
    hipLaunchKernelGGL(k_f32_init_residual, dim3(g), dim3(kBlock), 0, stream, n, bf, t, r, partial);
    AVS_TRY(reduce_stage(w, g, 2, OP_INIT, tol, 0, stream, nullptr));
    hipLaunchKernelGGL(k_f32_threshold, dim3(1), dim3(1), 0, stream, sc, tol);
    if (coded) hipLaunchKernelGGL(k_f32_init_p<true>, dim3(g), dim3(kBlock), 0, stream, n, r, invd, dcode, p, xf, partial, sc);
    else hipLaunchKernelGGL(k_f32_init_p<false>, dim3(g), dim3(kBlock), 0, stream, n, r, invd, dcode, p, xf, partial, sc);
    AVS_TRY(reduce_stage(w, g, 1, OP_RHO0, tol, 0, stream, nullptr));
    AVS_HIP(hipGetLastError());

    int enqueued = 0, last_chunk = 0;
    double spmv_ms_sum = 0.;
    int spmv_samples = 0;
    const bool sample = (info != nullptr);
    bool cancelled = false;
    bool timed_chunk = true;
    const bool use_graph = cur_opt().graph != 0;
    // float vectors are half as large: matrix + vectors fit the Infinity Cache more often (same rule, half the vector bytes)
    const int keep = A.keep_cached ? 1 : 0;
    auto enqueue_iteration = [&](int c, bool timed) -> avs_status {
        int nb = 0;
        if (timed) AVS_HIP(hipEventRecord(w->evA[c], stream));
        AVS_TRY(spmv_f32_dispatch<true>(A, p, t, partial, sc, stream, &nb)); // tmp = A p ; p.tmp
        if (timed) AVS_HIP(hipEventRecord(w->evB[c], stream));
        const int parity = c & 1; // (every chunk starts at an even iteration)
        const bool fuse_alpha = nb <= kFuseAlphaMax;
        double *vpart = partial + (w->npartial / 2); // (the SpMV's partials are still being read)
        if (fuse_alpha) {
            if (coded) AVS_F32_LAUNCH_R(true, true, n, r, t, invd, dcode, sc, vpart, partial, nb, parity);
            else AVS_F32_LAUNCH_R(false, true, n, r, t, invd, dcode, sc, vpart, partial, nb, parity);
        } else {
            AVS_TRY(reduce_stage(w, nb, 1, parity ? OP_ALPHA_ODD : OP_ALPHA, tol, 1, stream, nullptr));
            if (coded) AVS_F32_LAUNCH_R(true, false, n, r, t, invd, dcode, sc, vpart, (const double *)nullptr, 0, parity);
            else AVS_F32_LAUNCH_R(false, false, n, r, t, invd, dcode, sc, vpart, (const double *)nullptr, 0, parity);
        }
        if (coded) AVS_F32_LAUNCH_XP(true, n, xf, p, r, invd, dcode, sc, vpart, g, parity);
        else AVS_F32_LAUNCH_XP(false, n, xf, p, r, invd, dcode, sc, vpart, g, parity);
        return AVS_OK;
    };
    for (;;) {
        AVS_HIP(hipMemcpyAsync(w->host_sc, sc, sizeof(PcgScalars), hipMemcpyDeviceToHost, stream));
        AVS_HIP(hipStreamSynchronize(stream));
        if (sample && last_chunk > 0 && timed_chunk) {
            const int ran = w->host_sc->iter + ((w->host_sc->done == 1 || w->host_sc->done == 2) ? 1 : 0);
            const int first = enqueued - last_chunk;
            for (int c2 = 0; c2 < last_chunk && first + c2 < ran; c2 += kSampleEvery) {
                float ems = 0.f;
                if (hipEventElapsedTime(&ems, w->evA[c2], w->evB[c2]) == hipSuccess) {
                    spmv_ms_sum += ems;
                    ++spmv_samples;
                }
            }
        }
        if (w->host_sc->done || enqueued >= max_iters) break;
        if (cancel_consume()) { cancelled = true; break; }
        const int chunk = (max_iters - enqueued) < kChunk ? (max_iters - enqueued) : kChunk;
        const bool replay = use_graph && (enqueued / kChunk) % kTimedChunkEvery != 0 && chunk == kChunk && !w->graph_broken;
        timed_chunk = !replay;
        if (replay) {
            const void *key[10] = {A.row_ptr, A.col, A.val, A.codes, A.packed, A.table, xf, (const void *)(intptr_t)A.n,
                                   (const void *)(intptr_t)(A.table_size * 64 + A.col_bits),
                                   (const void *)(intptr_t)((coded ? 1 : 0) | 2 | (brick ? 4 : 0) | (int64_t)(A.epoch << 3))};
            if (w->graph && (memcmp(key, w->graph_key, sizeof(key)) != 0 || w->graph_tol != tol)) {
                (void)hipGraphExecDestroy(w->graph);
                w->graph = nullptr;
            }
            if (!w->graph) {
                hipGraph_t gr = nullptr;
                bool ok = hipStreamBeginCapture(stream, hipStreamCaptureModeRelaxed) == hipSuccess;
                if (ok) {
                    for (int c = 0; c < kChunk && ok; ++c) ok = enqueue_iteration(c, false) == AVS_OK;
                    ok = (hipStreamEndCapture(stream, &gr) == hipSuccess) && ok && gr;
                }
                if (ok) ok = hipGraphInstantiate(&w->graph, gr, nullptr, nullptr, 0) == hipSuccess;
                if (gr) (void)hipGraphDestroy(gr);
                if (!ok) {
                    (void)hipGetLastError();
                    w->graph = nullptr;
                    w->graph_broken = true;
                } else {
                    memcpy(w->graph_key, key, sizeof(key));
                    w->graph_tol = tol;
                }
            }
        }
        if (replay && w->graph) {
            AVS_HIP(hipGraphLaunch(w->graph, stream));
        } else {
            timed_chunk = true;
            for (int c = 0; c < chunk; ++c) AVS_TRY(enqueue_iteration(c, sample && (c % kSampleEvery == 0)));
        }
        AVS_HIP(hipGetLastError());
        enqueued += chunk;
        last_chunk = chunk;
    }
    hipLaunchKernelGGL(k_f32_widen, dim3(g), dim3(kBlock), 0, stream, n, (const float *)xf, x);
    AVS_HIP(hipEventRecord(w->ev1, stream));
    AVS_HIP(hipEventSynchronize(w->ev1));
    float ms = 0.f;
    AVS_HIP(hipEventElapsedTime(&ms, w->ev0, w->ev1));
    if (info) {
        const PcgScalars &h = *w->host_sc;
        info->iterations = h.iter;
        info->converged = (h.done != 0) ? 1 : 0;
        info->rhs_norm2 = h.rhs_norm2;
        info->error = (h.done == 3 || h.rhs_norm2 == 0.) ? 0. : (double)sqrtf((float)h.rr / (float)h.rhs_norm2);
        info->n = n;
        info->nnz = A.nnz;
        info->solve_ms = ms;
        info->spmv_ms = spmv_samples ? spmv_ms_sum / spmv_samples : 0.;
        info->resident = 0;
        info->cancelled = cancelled ? 1 : 0;
    }
    return AVS_OK;
}
#undef AVS_F32_LAUNCH_R
#undef AVS_F32_LAUNCH_XP

#ifdef AVS_PROBES
// probe / test entry: y = A x through the float forms (x holds float values; y is widened), + the folded partial sums of the fused dot
avs_status spmv_f32_probe(const CsrView &A, const double *x, double *y, bool fused, double *dot_out, hipStream_t st)
{
    const int64_t n = A.n;
    DevBuf<float> xf, yf;
    DevBuf<double> partial;
    AVS_TRY(xf.alloc((size_t)n + 8));
    AVS_TRY(yf.alloc((size_t)n + 8));
    const int g = stream_grid(n) < kVecGrid ? stream_grid(n) : kVecGrid;
    hipLaunchKernelGGL(k_f32_narrow, dim3(g), dim3(kBlock), 0, st, n, x, xf.p);
    if (!fused) {
        AVS_TRY(spmv_f32_dispatch<false>(A, xf.p, yf.p, nullptr, nullptr, st, nullptr));
    } else {
        size_t np = (size_t)stream_grid(n) + 16;
        if (A.brick && A.brick->ntiles > 0 && (size_t)brick_partial_count(*A.brick, 4) > np) np = (size_t)brick_partial_count(*A.brick, 4);
        AVS_TRY(partial.alloc(np));
        AVS_HIP(hipMemsetAsync(partial.p, 0, np * sizeof(double), st));
        AVS_TRY(spmv_f32_dispatch<true>(A, xf.p, yf.p, partial.p, nullptr, st, nullptr));
        if (dot_out) {
            std::vector<double> h(np);
            AVS_HIP(hipMemcpyAsync(h.data(), partial.p, np * sizeof(double), hipMemcpyDeviceToHost, st));
            AVS_HIP(hipStreamSynchronize(st));
            double s = 0.;
            for (double v : h) s += v;
            *dot_out = s;
        }
    }
    hipLaunchKernelGGL(k_f32_widen, dim3(g), dim3(kBlock), 0, st, n, (const float *)yf.p, y);
    AVS_HIP(hipGetLastError());
    AVS_HIP(hipStreamSynchronize(st));
    return AVS_OK;
}
#endif // AVS_PROBES
