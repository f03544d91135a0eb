cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
AVS_TRACE_PHASES=1 timeout 900 python tools/probes/slab_time.py 512 4 8 0 beam > gpurun_out/r06_slab_trace.log 2>&1; grep -n "avs " gpurun_out/r06_slab_trace.log | tail -70
