cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 700 python tools/probes/slab_time.py 1024 5 8 0 sheet > gpurun_out/r06_slab_time_sheet1024_w8.log 2>&1; tail -3 gpurun_out/r06_slab_time_sheet1024_w8.log | cut -c1-900
