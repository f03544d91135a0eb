# round 6, call 7: in-loop A/B of the priority placements and the non-temporal streams (product builds, one box, alternating)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B="python bench.py --no-extra --no-cpu-baseline --steps 3 --warmup 1"
E=$GRAFT_REPO_ROOT/adaptiveviscositysolver_amd/exp
for i in 1 2 3; do
timeout 600 $B > gpurun_out/r06_ab7_default_$i.log 2>&1
for v in prio0 prio1 prio2 pnt nornt; do
AVS_LIB_PATH=$E/libavs_hip_$v.so timeout 600 $B > gpurun_out/r06_ab7_${v}_$i.log 2>&1
done
done
for i in 1 2; do
timeout 600 $B --config 5 > gpurun_out/r06_ab7c5_default_$i.log 2>&1
AVS_LIB_PATH=$E/libavs_hip_prio0.so timeout 600 $B --config 5 > gpurun_out/r06_ab7c5_prio0_$i.log 2>&1
AVS_LIB_PATH=$E/libavs_hip_prio2.so timeout 600 $B --config 5 > gpurun_out/r06_ab7c5_prio2_$i.log 2>&1
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06_ab7*.log')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d['value'], 'spmv us', d['roofline']['mean_launch_us'], 'iters', d['config']['cg_iterations_per_step'])
    except Exception as e:
        print(f, 'ERR', e)
PY
