# A/B of the whole PCG loop on ONE box: reference product library (AVS_LIB_PATH, built from an earlier commit) against the current one
R=$GRAFT_REPO_ROOT
for rep in 1 2; do for lib in libavs_hip_ref.so libavs_hip.so; do
  [ -f $R/adaptiveviscositysolver_amd/$lib ] || continue
  echo "== $lib"; AVS_LIB_PATH=$R/adaptiveviscositysolver_amd/$lib python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('it/s %.0f  ms/step %.1f  SpMV us %.1f  iterations %d' % (d['value'], d['ms_per_step'], d['roofline'].get('mean_launch_us', 0), d['config']['cg_iterations_per_step']))"
done; done
