# round 6, call 6: priorities (three alternations), fused vector update tests + A/B, non-temporal r / y variants (A/B on one box)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python tools/probes/prio_probe.py 512 > gpurun_out/r06_prio_abc.log 2>&1
timeout 1500 python -m pytest tests/test_gpu_fused_vectors.py -x -q > gpurun_out/r06_t6.log 2>&1; tail -5 gpurun_out/r06_t6.log
B="python bench.py --no-extra --no-cpu-baseline --steps 3 --warmup 1"
for i in 1 2; do
AVS_PCG_FUSE_VECTORS=0 timeout 600 $B > gpurun_out/r06_ab_base_$i.log 2>&1
AVS_PCG_FUSE_VECTORS=1 timeout 600 $B > gpurun_out/r06_ab_fused_$i.log 2>&1
for v in ynt rnt yrnt; do
AVS_LIB_PATH=$GRAFT_REPO_ROOT/adaptiveviscositysolver_amd/exp/libavs_hip_$v.so timeout 600 $B > gpurun_out/r06_ab_${v}_$i.log 2>&1
done
done
grep -v "rows differ" gpurun_out/r06_prio_abc.log | grep "us:"
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06_ab_*.log')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d['value'], 'spmv us', d['roofline']['mean_launch_us'], 'iters', d['config']['cg_iterations_per_step'])
    except Exception as e:
        print(f, 'ERR', e)
PY
