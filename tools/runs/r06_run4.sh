# round 6, call 4: priorities as the default, fused vector update -- tests, A/B inside the loop (headline), configs[2] with the lowered threshold
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_fused_vectors.py tests/test_gpu_brick.py -x -q > gpurun_out/r06_t4.log 2>&1; tail -5 gpurun_out/r06_t4.log
timeout 600 python tools/probes/prio_probe.py 512 > gpurun_out/r06_prio_default.log 2>&1
for i in 1 2; do
AVS_PCG_FUSE_VECTORS=0 timeout 600 python bench.py --no-extra --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/r06_head_f0_$i.log 2>&1
AVS_PCG_FUSE_VECTORS=1 timeout 600 python bench.py --no-extra --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/r06_head_f1_$i.log 2>&1
done
timeout 300 python bench.py --config 3 --variable-viscosity --no-extra --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/r06_c3b.log 2>&1
AVS_PCG_FUSE_VECTORS=1 timeout 300 python bench.py --config 3 --variable-viscosity --no-extra --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/r06_c3b_f1.log 2>&1
grep -v "rows differ" gpurun_out/r06_prio_default.log | tail -12
for f in gpurun_out/r06_head_f*.log gpurun_out/r06_c3b*.log; do echo $f; tail -1 $f | cut -c1-200; done
