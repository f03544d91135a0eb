# round 6, call 13: E tiles (one pass through lattice + block region, word loads of the next round under the gathers) against the previous commit; new AUTO rule test
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_brick.py -x -q > gpurun_out/r06_t13.log 2>&1; tail -5 gpurun_out/r06_t13.log
E=$GRAFT_REPO_ROOT/adaptiveviscositysolver_amd/exp
for i in 1 2 3; do
for sc in "tank 512" "beam 512" "beam_mu 512" "sheet 1024" "tank 256"; do
timeout 300 python tools/probes/tank_bench.py $sc 2>&1 | grep "it/s"
AVS_LIB_PATH=$E/libavs_hip_prev.so timeout 300 python tools/probes/tank_bench.py $sc 2>&1 | grep "it/s"
done
done > gpurun_out/r06_etile_ab.log 2>&1
sort gpurun_out/r06_etile_ab.log
SPMV_SCENE=tank AVS_BRICK_DEBUG=80 SPMV_REPEATS=2 timeout 600 python tools/probes/spmv_time.py 512 2>&1 | grep "brick phases" | tail -3
