# HBM-side traffic of the default SpMV kernel, per MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE in SEPARATE
# --pmc passes (kernel filter: counter collection serialises every profiled dispatch), plus the request-size split.
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT
run() { name=$1; shift; timeout 200 rocprofv3 --kernel-include-regex "spmv|k_update_r" --pmc "$@" --output-format csv -d $R/gpurun_out/trf_$name -o p -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/trf_$name.log 2>&1; echo "pass $name rc=$?"; }
run fetch FETCH_SIZE
run write WRITE_SIZE
run req TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_32B_sum
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/trf_stats -o ${1:-r01h} -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/trf_stats.log 2>&1
cd $R && rm -f gpurun_out/trf_stats/*kernel_trace.csv && python - <<'PY'
import csv, collections, glob, json
out = {}
for f in sorted(glob.glob('gpurun_out/trf_*/p_counter_collection.csv')):
    acc = collections.defaultdict(lambda:[0,0.0])
    for r in csv.DictReader(open(f)):
        k = (r['Kernel_Name'].split('(')[0][-70:], r['Counter_Name'])
        acc[k][0]+=1; acc[k][1]+=float(r['Counter_Value'])
    for k,v in sorted(acc.items()):
        out.setdefault(k[0], {})[k[1]] = {"dispatches": v[0], "mean": v[1]/v[0]}
        print("  %-72s %-26s n=%d mean=%.5g" % (k[0], k[1], v[0], v[1]/v[0]))
json.dump(out, open('gpurun_out/trf_summary.json','w'), indent=1)
PY
grep -h '"metric"' gpurun_out/trf_stats.log | cut -c1-300
