cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_slab.py tests/test_gpu_prepass.py -x -q > gpurun_out/r06_t27.log 2>&1; grep -E "passed|failed" gpurun_out/r06_t27.log; grep -E "Error|assert " gpurun_out/r06_t27.log | head -5
timeout 1500 python -m pytest tests/test_gpu_bench_one_device.py -x -q -k slab > gpurun_out/r06_t27b.log 2>&1; grep -E "passed|failed" gpurun_out/r06_t27b.log
python - <<'PY'
import json
d=json.load(open('bench_extra.json'))
print(json.dumps(d['dist'].get('slab_local'))[:1500])
PY
timeout 700 python tools/probes/slab_time.py 1024 5 8 0 sheet > gpurun_out/r06_slab_time_sheet1024_w8.log 2>&1; tail -1 gpurun_out/r06_slab_time_sheet1024_w8.log | cut -c1-900
