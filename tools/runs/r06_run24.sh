cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/r06pp && mkdir -p $O
cd $R
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o pp -- python tools/probes/prepass_only.py 1024 5 sheet 6 > $O/pp.log 2>&1; tail -7 $O/pp.log
f=$(ls $O/stats/*/pp_kernel_stats.csv $O/stats/pp_kernel_stats.csv 2>/dev/null | head -1); cp "$f" $O/pp_kernel_stats.csv
rm -rf $O/stats
python - <<'PY'
import csv
for r in list(csv.DictReader(open('gpurun_out/r06pp/pp_kernel_stats.csv')))[:32]:
    print(r['Name'][:70].ljust(70), r['Calls'].rjust(5), ('%.1f'%(float(r['AverageNs'])/1e3)).rjust(9), ('%.2f'%(float(r['TotalDurationNs'])/1e6/6)).rjust(8))
PY
