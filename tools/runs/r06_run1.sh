# round 6, call 1: cost probes of the brick kernel (fused direction update, s_setprio); BASELINE configs[2] with the brick form forced
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python tools/probes/xp_probe.py 512 > gpurun_out/r06_xp_probe.log 2>&1
echo "== config 3 auto" > gpurun_out/r06_c3.log
timeout 300 python bench.py --config 3 --variable-viscosity --no-extra --no-cpu-baseline --steps 3 --warmup 1 >> gpurun_out/r06_c3.log 2>&1
echo "== config 3 AVS_BRICK=1" >> gpurun_out/r06_c3.log
AVS_BRICK=1 timeout 300 python bench.py --config 3 --variable-viscosity --no-extra --no-cpu-baseline --steps 3 --warmup 1 >> gpurun_out/r06_c3.log 2>&1
tail -30 gpurun_out/r06_xp_probe.log
