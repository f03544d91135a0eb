# Kernel stats of one frame loop of the 1024^3 thin sheet (BASELINE configs[4]): two pre-pass runs, apply, two assemblies, 2 solves, 4 transfers (rocprofv3 --kernel-trace --stats)
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/c5prof && mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o c5 -- python $R/bench.py --config 5 --no-cpu-baseline --no-extra --steps 1 --warmup 1 > $O/stats.log 2>&1; echo "rc=$?"
f=$(ls $O/stats/*/c5_kernel_stats.csv $O/stats/c5_kernel_stats.csv 2>/dev/null | head -1); cp "$f" $O/c5_kernel_stats.csv
rm -rf $O/stats
python - <<'PY'
import csv,os
rows=list(csv.DictReader(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/c5prof/c5_kernel_stats.csv')))
for r in rows[:60]:
    n=r['Name'].split('(')[0][-60:]
    print(f"{n:60s} calls {r['Calls']:>6s} total_ms {float(r['TotalDurationNs'])/1e6:9.2f} avg_us {float(r['AverageNs'])/1e3:10.1f}")
PY
