# a scene's tiles in k_spmv_brick (form forced): tile-walk variants (0 XCD-contiguous ranges, 32 tile = b + k grid), phase switches, stamps
R=$GRAFT_REPO_ROOT
export AVS_BRICK=1 SPMV_SCENE=${SPMV_SCENE:-sheet} SPMV_REPEATS=50
for d in 0 32 256 ${EXTRA_DEBUG}; do echo "== AVS_BRICK_DEBUG=$d"; AVS_BRICK_DEBUG=$d python $R/tools/probes/spmv_time.py ${1:-512} 2>&1 | grep -E "us:|rror|tiles" | sed -n '4,7p;9p'; done
echo "== stamps"; AVS_BRICK_DEBUG=80 SPMV_REPEATS=2 python $R/tools/probes/spmv_time.py ${1:-512} 2>&1 | grep "brick phases" | tail -1
