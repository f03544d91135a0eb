cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_slab.py -x -q -k "at_512" > gpurun_out/r06_t28.log 2>&1; tail -15 gpurun_out/r06_t28.log | cut -c1-400
