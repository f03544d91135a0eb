# round 6, call 10: four-lane fill runs as the default -- brick / float / fused tests; in-loop A/B against 16-lane runs on four workloads; register batches; 2-lane runs
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_brick.py tests/test_gpu_f32_loop.py tests/test_gpu_fused_vectors.py -x -q > gpurun_out/r06_t10.log 2>&1; tail -5 gpurun_out/r06_t10.log
B="python bench.py --no-extra --no-cpu-baseline --steps 3 --warmup 1"
E=$GRAFT_REPO_ROOT/adaptiveviscositysolver_amd/exp
for i in 1 2 3; do
timeout 600 $B > gpurun_out/r06_ab10_default_$i.log 2>&1
for v in rl4f2 rl2 rl16 rl4t4; do
AVS_LIB_PATH=$E/libavs_hip_$v.so timeout 600 $B > gpurun_out/r06_ab10_${v}_$i.log 2>&1
done
done
for i in 1 2; do
for c in "--config 5" "--config 3 --variable-viscosity" "--variable-viscosity" "--precision f32"; do
t=$(echo $c | tr -d ' -')
timeout 600 $B $c > gpurun_out/r06_ab10w_${t}_default_$i.log 2>&1
AVS_LIB_PATH=$E/libavs_hip_rl16.so timeout 600 $B $c > gpurun_out/r06_ab10w_${t}_rl16_$i.log 2>&1
done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06_ab10*.log')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d['value'], 'spmv us', d['roofline']['mean_launch_us'], 'iters', d['config']['cg_iterations_per_step'], 'asm', d.get('assembly_wall_ms'))
    except Exception as e:
        print(f, 'ERR', e, open(f).read()[-300:])
PY
