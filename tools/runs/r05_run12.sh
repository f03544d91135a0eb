cd $GRAFT_REPO_ROOT && O=gpurun_out/r05j && mkdir -p $O
timeout 1200 python tools/loopback_scaling.py --n 512 --levels 4 --worlds 1,2,4,8 > $O/r05_loopback_scaling.json 2> $O/loopback.err; echo "loopback rc=$?"
python - <<'PY'
import json
d = json.load(open('gpurun_out/r05j/r05_loopback_scaling.json'))
for w, v in d['worlds'].items():
    print(w, 'max us/it %.1f' % v['max_us_per_iteration'], 'ranks', ['%.1f' % r['us_per_iteration'] for r in v['ranks']], [r['n_own'] for r in v['ranks']], 'e2e ms %.1f' % v['projected_end_to_end_ms_1271_iterations'])
PY
timeout 1500 python -m pytest tests/test_gpu_dist.py tests/test_gpu_bench_one_device.py -x -q -m gpu > $O/dist.log 2>&1; echo "dist rc=$?"; grep -E "passed|failed|Error" $O/dist.log | tail -3
