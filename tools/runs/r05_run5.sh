# planned walk with a dynamic tail: correctness, then strided / planned / planned + drawn tail on three scenes, then the loops
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/r05e && mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_brick.py tests/test_gpu_cancel.py -x -q -m gpu > $O/brick_cancel.log 2>&1; echo "brick+cancel rc=$?"; tail -3 $O/brick_cancel.log
timeout 900 python -m pytest tests/test_gpu_dist.py -x -q -m gpu -k "brick or local_brick or rccl_world or loopback" > $O/dist_brick.log 2>&1; echo "dist brick rc=$?"; tail -3 $O/dist_brick.log
T=$R/tools/probes/spmv_time.py
{
for scene in beam sheet tank; do
  for cfg in "0 0" "1 0" "1 0.1" "1 0.2"; do
    set -- $cfg
    echo "== $scene AVS_BRICK_PLAN=$1 AVS_BRICK_DYN=$2"; SPMV_SCENE=$scene AVS_BRICK_PLAN=$1 AVS_BRICK_DYN=$2 SPMV_REPEATS=100 timeout 200 python $T 512 2>&1 | grep -E "fused-dot|default SpMV|rror" | grep -v stream
  done
done
} 2>&1 | tee $O/walks.log
echo "== PCG loop A/B (headline)"; bash tools/probes/loop_ab.sh 2>&1 | tee $O/ab_loop.log
echo "== force-dist (direct loop, world 1)"; bash tools/probes/loop_ab.sh --force-dist 2>&1 | tee $O/ab_loop_forcedist.log
