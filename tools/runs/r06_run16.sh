# round 6, call 16: slab-local timing, each rank alone on the GPU
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python tools/probes/slab_time.py 512 4 8 0 beam > gpurun_out/r06_slab_time_beam512_w8.log 2>&1; tail -12 gpurun_out/r06_slab_time_beam512_w8.log | cut -c1-900
timeout 600 python tools/probes/slab_time.py 512 4 2 0 beam > gpurun_out/r06_slab_time_beam512_w2.log 2>&1; tail -1 gpurun_out/r06_slab_time_beam512_w2.log | cut -c1-900
