# k_spmv_brick COLD (600 MB overwritten between launches: what the vector kernels of a PCG iteration do to the caches) against warm,
# with the tile-walk variants (AVS_BRICK_DEBUG: 0 XCD-contiguous strided, 32 tile = b + k grid, 128 consecutive tiles per workgroup)
R=$GRAFT_REPO_ROOT
for d in 0 32 128; do for th in 0 600; do
  echo "== AVS_BRICK_DEBUG=$d thrash=${th}MB"; AVS_BRICK_DEBUG=$d AVS_BENCH_THRASH_MB=$th SPMV_REPEATS=100 python $R/tools/probes/spmv_time.py ${1:-512} 2>&1 | grep -E "fused-dot|rror" | tail -3
done; done
