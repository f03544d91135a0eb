# round 6, call 9: fill-run statistics; in-loop A/B: tail size of the priorities, graded priorities, fill runs of 8 / 4 lanes
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python tools/probes/wave_stats.py 512 > gpurun_out/r06_wave_stats_beam.log 2>&1
SPMV_SCENE=sheet timeout 600 python tools/probes/wave_stats.py 1024 > gpurun_out/r06_wave_stats_sheet.log 2>&1
grep -h "brick\|tiles\|histogram" gpurun_out/r06_wave_stats_*.log
B="python bench.py --no-extra --no-cpu-baseline --steps 3 --warmup 1"
E=$GRAFT_REPO_ROOT/adaptiveviscositysolver_amd/exp
for i in 1 2 3; do
timeout 600 $B > gpurun_out/r06_ab9_default_$i.log 2>&1
for v in tail3 tail4 tail6 graded rl8 rl4; do
AVS_LIB_PATH=$E/libavs_hip_$v.so timeout 600 $B > gpurun_out/r06_ab9_${v}_$i.log 2>&1
done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r06_ab9*.log')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d['value'], 'spmv us', d['roofline']['mean_launch_us'], 'iters', d['config']['cg_iterations_per_step'], 'asm', d.get('assembly_wall_ms'))
    except Exception as e:
        print(f, 'ERR', e, open(f).read()[-300:])
PY
