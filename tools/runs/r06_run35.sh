cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/r06tr && mkdir -p $O
cd $R
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o tr -- python tools/probes/slab_transfer_prof.py 512 4 beam 3 > $O/tr.log 2>&1; grep transfer $O/tr.log
f=$(ls $O/stats/*/tr_kernel_stats.csv $O/stats/tr_kernel_stats.csv 2>/dev/null | head -1); cp "$f" $O/tr_kernel_stats.csv; rm -rf $O/stats
python - <<'PY'
import csv
for r in list(csv.DictReader(open('gpurun_out/r06tr/tr_kernel_stats.csv'))):
    if any(k in r['Name'] for k in ('nodes','apply_regular','scatter','ridx','fillBuffer','copyBuffer')):
        print(r['Name'][:64].ljust(64), r['Calls'].rjust(5), ('%.1f'%(float(r['AverageNs'])/1e3)).rjust(9), ('%.2f'%(float(r['TotalDurationNs'])/1e6)).rjust(8))
PY
