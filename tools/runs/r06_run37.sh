cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_prepass.py tests/test_gpu_slab.py tests/test_gpu_post.py tests/test_scene_equivalents.py tests/test_gpu_parity.py -x -q > gpurun_out/r06_t37.log 2>&1; grep -E "passed|failed" gpurun_out/r06_t37.log; grep -E "Error|assert " gpurun_out/r06_t37.log | head -5
timeout 300 python tools/probes/prepass_only.py 512 4 beam 5 | tail -2
timeout 300 python tools/probes/prepass_only.py 1024 5 sheet 5 | tail -2
timeout 300 python bench.py --scene beam --no-cpu-baseline --no-extra --steps 2 --warmup 1 > /dev/null 2>&1; python - <<'PY'
import json
d=json.load(open('bench_extra.json'))
print({k:d.get(k) for k in ('value','prepass_ms','ms_per_step','assembly_ms')})
PY
