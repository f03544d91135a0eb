# k_spmv_brick on the DEVICE-built form with phases switched off (AVS_BRICK_DEBUG: 1 no fill, 2 no pattern rows, 4 no streamed rows,
# 32 tile = b + k gridDim instead of XCD-contiguous ranges, 16|64 phase stamps printed per launch)
R=$GRAFT_REPO_ROOT
for d in 0 32 1 2 4 7; do echo "== AVS_BRICK_DEBUG=$d"; AVS_BRICK_DEBUG=$d python $R/tools/probes/spmv_time.py ${1:-512} 2>&1 | grep -E "fused-dot|default SpMV|rror" | tail -4; done
echo "== stamps"; AVS_BRICK_DEBUG=80 SPMV_REPEATS=2 python $R/tools/probes/spmv_time.py ${1:-512} 2>&1 | grep "brick phases" | tail -4
