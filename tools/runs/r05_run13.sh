cd $GRAFT_REPO_ROOT && O=gpurun_out/r05k && mkdir -p $O
python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.load(open('gpurun_out/r05k/bench_default.json'))
print('headline', round(d['value']), d['roofline']['mean_launch_us'], d['roofline']['frac'])
for e in d['extra_workloads']:
    if 'error' in e: print(e); continue
    print(e['workload'][:60], '| it/s %.0f' % e['value'], '| ms/step %.2f first %.2f new-matrix %.2f' % (e['ms_per_step'], e['first_solve_ms'], e['new_matrix_solve_ms']), '| resident', e['resident_loop'], '| spmv us', (e['roofline'] or {}).get('mean_launch_us'), '| asm', round(e['assembly_wall_ms'], 1))
PY
AVS_CG_RESIDENT_VERBOSE=1 python bench.py --scene beam --no-cpu-baseline --no-extra 2>&1 | grep "avs resident" | head -30
