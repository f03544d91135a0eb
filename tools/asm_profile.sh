#!/bin/bash
# per-kernel times of ONE assembly pass on the headline workload (rocprofv3 kernel trace of bench.py with 8 PCG iterations)
out=${1:-gpurun_out/asm_prof}
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out -o a -- python $root/bench.py --steps 1 --warmup 0 --max-iters 8 --no-cpu-baseline > /dev/null 2>&1
cd $root
python - <<PY
import csv,glob
f=glob.glob("$out/**/a_kernel_stats.csv", recursive=True)[0]
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
for r in rows[:45]:
    print("%-70s calls %5s avg_us %9.1f tot_ms %8.2f"%(r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6))
PY
