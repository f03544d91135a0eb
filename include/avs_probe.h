/* avs_probe.h -- measurement and test entries of libavs_hip.so that are NOT part of the plugin seam (include/avs.h).
 * tools/, tests/ and bench.py's probes bind these; a Houdini host never does. */
#ifndef AVS_PROBE_H
#define AVS_PROBE_H
#include "avs.h"
#ifdef __cplusplus
extern "C" {
#endif

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
