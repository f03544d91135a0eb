cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_prepass.py tests/test_gpu_slab.py tests/test_gpu_post.py tests/test_scene_equivalents.py tests/test_gpu_parity.py -x -q > gpurun_out/r06_t25.log 2>&1; grep -E "passed|failed" gpurun_out/r06_t25.log; tail -30 gpurun_out/r06_t25.log | grep -E "Error|assert" | head
bash tools/runs/r06_run24.sh 2>&1 | tail -34
timeout 300 python tools/probes/prepass_only.py 512 4 beam 5
