cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/r06sc && mkdir -p $O
cd $R
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o sc -- python bench.py --scene beam --no-cpu-baseline --no-extra --steps 1 --warmup 1 > $O/sc.log 2>&1
f=$(ls $O/stats/*/sc_kernel_stats.csv $O/stats/sc_kernel_stats.csv 2>/dev/null | head -1); cp "$f" $O/sc_kernel_stats.csv; rm -rf $O/stats
python - <<'PY'
import csv
for r in list(csv.DictReader(open('gpurun_out/r06sc/sc_kernel_stats.csv'))):
    if any(k in r['Name'] for k in ('k_rows','merge_rows','unique_rows','wave_slots','scan','permute','vi_','bk_','cwin','tlt','invert','brick_keys','Radix','radix','resident')):
        print(r['Name'][:70].ljust(70), r['Calls'].rjust(5), ('%.1f'%(float(r['AverageNs'])/1e3)).rjust(9))
PY
