# round 6, call 8: second in-loop A/B (tail size / levels of the priorities, non-temporal row descriptors, streamed phase priority)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B="python bench.py --no-extra --no-cpu-baseline --steps 3 --warmup 1"
E=$GRAFT_REPO_ROOT/adaptiveviscositysolver_amd/exp
for i in 1 2 3; do
timeout 600 $B > gpurun_out/r06_ab8_default_$i.log 2>&1
for v in tail1 tail3 tp2 mnt sprio; do
AVS_LIB_PATH=$E/libavs_hip_$v.so timeout 600 $B > gpurun_out/r06_ab8_${v}_$i.log 2>&1
done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06_ab8*.log')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d['value'], 'spmv us', d['roofline']['mean_launch_us'], 'iters', d['config']['cg_iterations_per_step'])
    except Exception as e:
        print(f, 'ERR', e)
PY
