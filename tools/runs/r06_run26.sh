cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_prepass.py tests/test_gpu_slab.py tests/test_gpu_post.py tests/test_scene_equivalents.py -x -q > gpurun_out/r06_t26.log 2>&1; grep -E "passed|failed" gpurun_out/r06_t26.log
timeout 600 python bench.py --config 5 --no-cpu-baseline --no-extra --steps 2 --warmup 1 > gpurun_out/r06_c5.log 2>&1; python - <<'PY'
import json
d=json.load(open('bench_extra.json'))
print({k:d.get(k) for k in ('value','prepass_ms','prepass_apply_ms','transfer_to_regular_grid_ms','transfer_in_place_ms','assembly_ms','end_to_end_ms')})
PY
timeout 900 python tools/probes/slab_time.py 512 4 8 0 beam > gpurun_out/r06_slab_time_beam512_w8.log 2>&1; tail -1 gpurun_out/r06_slab_time_beam512_w8.log | cut -c1-700
timeout 700 python tools/probes/slab_time.py 1024 5 8 0 sheet > gpurun_out/r06_slab_time_sheet1024_w8.log 2>&1; tail -1 gpurun_out/r06_slab_time_sheet1024_w8.log | cut -c1-900
