# A/B of the brick kernel on ONE box: reference probe library (AVS_PROBE_LIB_PATH, built from an earlier commit) against the current one
R=$GRAFT_REPO_ROOT
for lib in libavs_probe_ref.so libavs_probe.so; do
  [ -f $R/adaptiveviscositysolver_amd/$lib ] || continue
  echo "== $lib"; AVS_PROBE_LIB_PATH=$R/adaptiveviscositysolver_amd/$lib python $R/tools/probes/spmv_time.py ${1:-512} 2>&1 | grep -E "us:|tiles|rror"
done
