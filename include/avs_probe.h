/* avs_probe.h -- measurement and test entries that are NOT part of the plugin seam (include/avs.h).  They are compiled (-DAVS_PROBES) into
 * libavs_probe.so only -- a superset build of libavs_hip.so from the same sources that also carries the SpMV kernel sweeps, the SELL and
 * stream probes and the fault-injection hooks (AVS_DIST_INJECT_STALE, AVS_CG_RESIDENT_FAKE_FAULT).  tools/ and the tests that need them
 * load that library; the product library exports none of this. */
#ifndef AVS_PROBE_H
#define AVS_PROBE_H
#include "avs.h"
#ifdef __cplusplus
extern "C" {
#endif

/* One SpMV y = A x on device-resident CSR (measurement entry: the graded kernel).
 * `variant` selects the kernel (0 = library default).  Enqueues `repeats` launches. */
avs_status avs_spmv_csr(int64_t n, const int32_t *row_ptr, const int32_t *col, const double *val,
                        const double *x, double *y, int32_t variant, int32_t repeats, void *stream);
/* Measurement entry for the SELL-C-sigma experiment (C = 64: one wavefront per slice; BASELINE configs[4]): slice s holds 64
 * consecutive rows column-major, entry j of lane l at slice_ptr[s] + 64 j + l, padded with (col 0, val 0.0); device pointers.
 * y comes out in the slice (sigma-sorted) row order.  tools/sell_experiment.py builds the layout. */
avs_status avs_spmv_sell(int64_t nslices, const int64_t *slice_ptr, const int32_t *col, const double *val, const double *x,
                         double *y, int32_t repeats, void *stream, double *ms_per_launch);
/* SpMV on the system owned by ctx (after avs_assemble), same kernel the solver uses;
 * returns the mean HIP-event time per launch in *ms_per_launch. */
avs_status avs_bench_spmv(avs_ctx *ctx, int32_t variant, int32_t repeats, double *ms_per_launch);

/* y = A x with the storage form and kernel the solver's loop launches for the system owned by ctx (brick-structured form, word stream,
 * ...), for an ARBITRARY x: x and y are device vectors in the REFERENCE's DOF numbering (the entry permutes into the solver's brick-major
 * numbering and back).  fused_dot != 0 launches the fused-dot instantiation (what the PCG loop runs) and returns the folded x.y in
 * *dot_out.  The parity tests compare y with the CPU oracle's CSR product bit for bit. */
avs_status avs_spmv_solver_form(avs_ctx *ctx, const double *x, double *y, int32_t fused_dot, double *dot_out);
/* the same for the LOCAL system of a partitioned solve (after avs_dist_assemble / avs_dist_partition): x_ext holds the rank's
 * [owned | halo] entries in local numbering (n_own + n_halo doubles, device), y its n_own rows */
avs_status avs_dist_spmv_local_form(avs_ctx *ctx, const double *x_ext, double *y, int32_t fused_dot, double *dot_out);

/* Measurement: load balance of the brick kernel's row walk -- per G tile the quads of the slowest of the eight waves against the mean wave
 * (printed to stderr; out6 = {tiles, rows per tile, quads per row, slowest-wave quads, mean-wave quads, 0}) */
avs_status avs_brick_wave_stats(avs_ctx *ctx, double *out6);

/* Measured stream ceilings of the device for the access pattern of the SpMV's matrix stream
 * (mode 0: read-only 16 B/lane, 1: read-only non-temporal, 2: copy); GB/s of bytes moved. */
avs_status avs_bench_stream(int32_t mode, int64_t bytes, int32_t repeats, int32_t device, double *gbps);

/* The brick-structured SpMV form (csrc/avs_brick.hip) as plain device arrays, for a form built OUTSIDE the library
 * (tools/brick_build.py, the reference builder the device builder is tested against). */
typedef struct {
    int32_t ntiles;
    const uint32_t *tile_blk; /* 2 x uint32 per tile: first 16-B unit of its descriptor block, units */
    const uint32_t *blocks;   /* descriptor blocks (layout: csrc/avs_brick.hip) */
    const uint32_t *rdesc;    /* 2 x uint32 per pattern row (execution order): descriptor, position in the tile */
    const uint16_t *ownslot;  /* per row: own lattice slot in its tile, 0xffff none */
    const uint32_t *pwords;   /* global pattern words */
    const uint32_t *sdesc;    /* 2 x uint32 per streamed row: local row | len << 16, first word relative to the tile's */
    const uint32_t *swords;   /* streamed words (code << col_bits | column), CSR order */
    const double *table;      /* value dictionary */
    int32_t table_size, col_bits;
} avs_brick_arrays;
/* y = A x (+ per-wave partials of x.y when partial != NULL) with the brick kernel, `repeats` timed launches */
avs_status avs_brick_spmv_probe(const avs_brick_arrays *a, const double *x, double *y, double *partial, int32_t repeats, void *stream,
                                double *ms_per_launch);

#ifdef __cplusplus
}
#endif
#endif
