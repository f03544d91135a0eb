cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2700 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_bench_one_device.py > gpurun_out/r06_t22_full.log 2>&1; tail -15 gpurun_out/r06_t22_full.log
