#!/bin/bash
# per-kernel times of the PCG loop on the headline workload (rocprofv3 kernel stats of one bench step)
out=${1:-gpurun_out/loop_prof}
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out -o a -- python $root/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $root/$out.log 2>&1
cd $root
grep -h '"metric"' $out.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('it/s', round(d['value']), 'ms/step', round(d['ms_per_step'],2), 'frac', round(d['roofline']['frac'],4))"
python - <<PY
import csv,glob
f=glob.glob("$out/**/a_kernel_stats.csv", recursive=True)[0]
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
for r in rows[:6]:
    print("%-60s calls %5s avg_us %8.2f"%(r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3))
PY
