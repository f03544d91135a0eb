# round 6: evidence on the final tree -- counters first (the bench then quotes them: same fingerprint), then the default bench, then the f32 and 1024^3 records
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/profile_r06.sh r06 f64 > gpurun_out/r06_prof_f64.log 2>&1; tail -3 gpurun_out/r06_prof_f64.log | cut -c1-300
cd $GRAFT_REPO_ROOT; cp gpurun_out/r06prof/r06_spmv_counters.json gpurun_out/r06prof/r06_spmv_counters_f64.json; cp gpurun_out/r06prof/r06_kernel_stats.csv gpurun_out/r06prof/r06_kernel_stats_f64.csv; cp gpurun_out/r06prof/r06_pmc_summary.json gpurun_out/r06prof/r06_pmc_summary_f64.json
bash tools/profile_r06.sh r06f32 f32 > gpurun_out/r06_prof_f32.log 2>&1; tail -3 gpurun_out/r06_prof_f32.log | cut -c1-300
cd $GRAFT_REPO_ROOT; bash tools/profile_r06_resident.sh > gpurun_out/r06_prof_resident.log 2>&1; tail -4 gpurun_out/r06_prof_resident.log | cut -c1-300
cd $GRAFT_REPO_ROOT; python - <<'PY'
import json
a=[r for r in json.load(open('gpurun_out/r06prof/r06_spmv_counters_f64.json'))]+[r for r in json.load(open('gpurun_out/r06prof/r06f32_spmv_counters.json'))]
json.dump(a,open('profiles/spmv_counters.json','w'),indent=1)
import shutil; shutil.copy('gpurun_out/r06prof/resident_counters.json','profiles/resident_counters.json')
PY
cp profiles/spmv_counters.json gpurun_out/r06prof/spmv_counters_merged.json
python bench.py > gpurun_out/r06_final_bench.log 2>&1; tail -1 gpurun_out/r06_final_bench.log | cut -c1-2600; cp bench_extra.json gpurun_out/r06_final_bench_extra.json
bash tools/profile_c5.sh > gpurun_out/r06_prof_c5.log 2>&1; tail -5 gpurun_out/r06_prof_c5.log | cut -c1-200
