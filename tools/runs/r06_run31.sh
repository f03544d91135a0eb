cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python tools/probes/slab_stress.py 1 16 > gpurun_out/r06_slab_stress_1.log 2>&1; grep -E "^ok|^FAIL|failures" gpurun_out/r06_slab_stress_1.log | cut -c1-400
