cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for seed in 2 3 4; do timeout 1200 python tools/probes/slab_stress.py $seed 14 > gpurun_out/r06_slab_stress_$seed.log 2>&1; grep -E "^ok|^FAIL|failures" gpurun_out/r06_slab_stress_$seed.log | cut -c1-300; done
