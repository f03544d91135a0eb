cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/sceneprof && mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o sc -- python $R/tools/probes/resident_plan_stages.py beam > $O/stats.log 2>&1; echo "rc=$?"
f=$(ls $O/stats/*/sc_kernel_stats.csv $O/stats/sc_kernel_stats.csv 2>/dev/null | head -1); cp "$f" $O/sc_kernel_stats.csv
python - <<'PY'
import csv,os
rows=list(csv.DictReader(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/sceneprof/sc_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows); calls=sum(int(r['Calls']) for r in rows)
print("kernels total ms", tot/1e6, "launches", calls)
for r in rows[:40]:
    n=r['Name'].split('(')[0][-46:]
    print(f"{n:46s} calls {r['Calls']:>5s} total_ms {float(r['TotalDurationNs'])/1e6:8.3f} avg_us {float(r['AverageNs'])/1e3:9.1f}")
PY
rm -rf $O/stats
