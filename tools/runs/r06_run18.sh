cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_slab.py tests/test_gpu_prepass.py -x -q > gpurun_out/r06_t18.log 2>&1; grep -E "passed|failed" gpurun_out/r06_t18.log
AVS_TRACE_PHASES=1 timeout 900 python tools/probes/slab_time.py 512 4 8 0 beam > gpurun_out/r06_slab_trace.log 2>&1; grep -n "avs " gpurun_out/r06_slab_trace.log | tail -22; tail -1 gpurun_out/r06_slab_trace.log
timeout 900 python tools/probes/slab_time.py 512 4 8 0 beam > gpurun_out/r06_slab_time_beam512_w8.log 2>&1; tail -9 gpurun_out/r06_slab_time_beam512_w8.log | cut -c1-700
