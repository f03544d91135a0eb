# round 6, call 3: priorities as the default -- brick tests, A/B inside the loop (headline), configs[2] with the lowered threshold
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_brick.py tests/test_gpu_f32_loop.py -x -q > gpurun_out/r06_t3.log 2>&1; tail -3 gpurun_out/r06_t3.log
timeout 600 python tools/probes/prio_probe.py 512 > gpurun_out/r06_prio_default.log 2>&1
for i in 1 2; do
timeout 600 python bench.py --no-extra --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/r06_head_$i.log 2>&1
done
timeout 300 python bench.py --config 3 --variable-viscosity --no-extra --no-cpu-baseline --steps 3 --warmup 1 > gpurun_out/r06_c3b.log 2>&1
grep -v "rows differ" gpurun_out/r06_prio_default.log | tail -12
for f in gpurun_out/r06_head_1.log gpurun_out/r06_head_2.log gpurun_out/r06_c3b.log; do tail -1 $f | cut -c1-330; done
