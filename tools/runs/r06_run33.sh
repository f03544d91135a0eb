cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_slab.py -x -q -k "transfer" > gpurun_out/r06_t33.log 2>&1; tail -25 gpurun_out/r06_t33.log | cut -c1-400
