# v7 brick kernel (8-B absolute-column runs, one round trip per tile): correctness, then A/B against the previous commit's libraries
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05b; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_brick.py tests/test_gpu_cancel.py -x -q -m gpu > $O/brick_cancel.log 2>&1; echo "brick+cancel rc=$?"; tail -3 $O/brick_cancel.log
timeout 600 python -m pytest tests/test_gpu_dist.py -x -q -m gpu -k "brick or local_brick" > $O/dist_brick.log 2>&1; echo "dist brick rc=$?"; tail -3 $O/dist_brick.log
for scene in beam sheet tank; do
  echo "== stand-alone SpMV, $scene"; SPMV_SCENE=$scene bash tools/probes/brick_ab.sh 512 2>&1 | tee -a $O/ab_spmv_$scene.log
done
echo "== PCG loop A/B (headline)"; bash tools/probes/loop_ab.sh 2>&1 | tee $O/ab_loop.log
echo "== force-dist (direct loop, world 1)"; bash tools/probes/loop_ab.sh --force-dist 2>&1 | tee $O/ab_loop_forcedist.log
