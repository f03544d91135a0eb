cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for seed in 5 6 7; do timeout 1200 python tools/probes/slab_stress.py $seed 14 > gpurun_out/r06_slab_stress_$seed.log 2>&1; grep -E "^FAIL|failures" gpurun_out/r06_slab_stress_$seed.log | cut -c1-300; grep -c "transfer equal" gpurun_out/r06_slab_stress_$seed.log; done
