# rocprofv3 kernel-trace + stats of the headline bench (no extras, no CPU leg): per-kernel times incl. the brick-form builder
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r04 -o ${1:-r04} -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $R/gpurun_out/prof_r04.log 2>&1
cd $R && rm -f gpurun_out/prof_r04/*kernel_trace.csv && grep -h '"metric"' gpurun_out/prof_r04.log | cut -c1-600
python - <<'PY'
import csv, glob
for f in glob.glob('gpurun_out/prof_r04/*kernel_stats.csv'):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r['TotalDurationNs']))
    for r in rows[:45]:
        print("%-90s calls=%6s avg_us=%10.1f total_ms=%9.2f %s%%" % (r['Name'][:90], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6, r['Percentage']))
PY
