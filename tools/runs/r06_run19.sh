cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_slab.py tests/test_gpu_prepass.py -x -q > gpurun_out/r06_t19.log 2>&1; grep -E "passed|failed" gpurun_out/r06_t19.log
timeout 900 python tools/probes/slab_time.py 512 4 8 0 beam > gpurun_out/r06_slab_time_beam512_w8.log 2>&1; tail -9 gpurun_out/r06_slab_time_beam512_w8.log | cut -c1-700
timeout 900 python tools/probes/slab_time.py 512 4 4 0 beam > gpurun_out/r06_slab_time_beam512_w4.log 2>&1; tail -1 gpurun_out/r06_slab_time_beam512_w4.log | cut -c1-700
timeout 900 python tools/probes/slab_time.py 512 4 2 0 beam > gpurun_out/r06_slab_time_beam512_w2.log 2>&1; tail -1 gpurun_out/r06_slab_time_beam512_w2.log | cut -c1-700
timeout 900 python tools/probes/slab_time.py 1024 5 8 0 sheet > gpurun_out/r06_slab_time_sheet1024_w8.log 2>&1; tail -3 gpurun_out/r06_slab_time_sheet1024_w8.log | cut -c1-900
