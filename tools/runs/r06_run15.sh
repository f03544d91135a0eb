# round 6, call 15: slab-local pre-pass + assembly, first GPU run
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_slab.py -q > gpurun_out/r06_t15.log 2>&1; tail -40 gpurun_out/r06_t15.log
timeout 900 python -m pytest tests/test_gpu_prepass.py tests/test_gpu_dist.py -x -q > gpurun_out/r06_t15b.log 2>&1; tail -5 gpurun_out/r06_t15b.log
