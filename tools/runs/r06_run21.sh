cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_bench_one_device.py -x -q > gpurun_out/r06_t21.log 2>&1; tail -15 gpurun_out/r06_t21.log
