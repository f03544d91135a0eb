cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_slab.py tests/test_gpu_post.py -x -q -k "transfer or post" > gpurun_out/r06_t36.log 2>&1; tail -3 gpurun_out/r06_t36.log | cut -c1-300
timeout 600 python tools/probes/slab_transfer_prof.py 512 4 beam 3 2>&1 | grep transfer | tail -2
timeout 600 python tools/probes/slab_time.py 512 4 8 0 beam > gpurun_out/r06_slab_time_beam512_w8.log 2>&1; tail -1 gpurun_out/r06_slab_time_beam512_w8.log | cut -c1-800
timeout 700 python tools/probes/slab_time.py 1024 5 8 0 sheet > gpurun_out/r06_slab_time_sheet1024_w8.log 2>&1; tail -1 gpurun_out/r06_slab_time_sheet1024_w8.log | cut -c1-800
