# per-rank iteration times of the 1/2/4/8-way partition (loop-back on one GPU) and the two partitioned loops at world 1, round-5 binary
cd $GRAFT_REPO_ROOT && O=gpurun_out/r05i && mkdir -p $O
timeout 1200 python tools/loopback_scaling.py --n 512 --levels 4 --worlds 1,2,4,8 > $O/r05_loopback_scaling.json 2> $O/loopback.err; echo "loopback rc=$?"
python - <<'PY'
import json
d = json.load(open('gpurun_out/r05i/r05_loopback_scaling.json'))
for w, v in d['worlds'].items():
    print(w, 'max us/it %.1f' % v['max_us_per_iteration'], 'ranks', ['%.1f' % r['us_per_iteration'] for r in v['ranks']], 'resident', [r['resident_loop'] for r in v['ranks']][:2], 'e2e ms %.1f' % v['projected_end_to_end_ms_1271_iterations'])
PY
for tr in rccl direct; do AVS_DIST_TRANSPORT=$tr python bench.py --force-dist --steps 3 --warmup 1 --no-cpu-baseline --no-extra 2>/dev/null > $O/forcedist_$tr.json; python -c "
import json; d = json.load(open('$O/forcedist_$tr.json')); print('$tr', 'it/s %.0f' % d['value'], 'SpMV us %.1f' % d['roofline']['mean_launch_us'], d['dist']['transport'], d['config']['cg_iterations_per_step'])"; done
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra 2>/dev/null > $O/headline.json; python -c "
import json; d = json.load(open('$O/headline.json')); print('standard', 'it/s %.0f' % d['value'], 'SpMV us %.1f' % d['roofline']['mean_launch_us'])"
