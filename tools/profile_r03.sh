# Round-3 profiles (run on the GPU box through gpurun; summaries are copied to profiles/ by hand):
#  1. rocprofv3 --kernel-trace --stats of the default bench (headline) and of the two scene-equivalents (CU-resident loop)
#  2. separate --pmc passes (FETCH_SIZE | WRITE_SIZE) for the SpMV of the cell-major system (the traffic figure bench.py quotes)
# MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE in separate passes; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B.
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/r03prof && mkdir -p $O
TAG=${1:-r03}
stats() { name=$1; shift; timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$name -o $TAG -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra "$@" > $O/stats_$name.log 2>&1; echo "stats $name rc=$?"; rm -f $O/stats_$name/*/*kernel_trace.csv $O/stats_$name/*kernel_trace.csv; }
stats uniform
stats beam --scene beam
stats buckling --scene buckling
pmc() { name=$1; shift; timeout 400 rocprofv3 --kernel-include-regex "spmv|k_update_r" --pmc "$@" --output-format csv -d $O/pmc_$name -o p -- python $R/bench.py --steps 1 --warmup 0 --max-iters 96 --no-cpu-baseline --no-extra > $O/pmc_$name.log 2>&1; echo "pmc $name rc=$?"; }
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
cd $R && python - <<'PY'
import csv, collections, glob, json
out = {}
for f in sorted(glob.glob('gpurun_out/r03prof/pmc_*/**/p_counter_collection.csv', recursive=True)):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        k = (r['Kernel_Name'].split('(')[0][-90:], r['Counter_Name'])
        acc[k][0] += 1
        acc[k][1] += float(r['Counter_Value'])
    for k, v in sorted(acc.items()):
        out.setdefault(k[0], {})[k[1]] = {"dispatches": v[0], "mean": v[1] / v[0]}
json.dump(out, open('gpurun_out/r03prof/pmc_summary.json', 'w'), indent=1)
print(json.dumps(out, indent=0)[:1500])
PY
grep -h '"metric"' $O/stats_*.log | cut -c1-260
for n in uniform beam buckling; do f=$(ls $O/stats_$n/*/${TAG}_kernel_stats.csv $O/stats_$n/${TAG}_kernel_stats.csv 2>/dev/null | head -1); echo "== $n"; head -6 "$f" | cut -c1-200; done
