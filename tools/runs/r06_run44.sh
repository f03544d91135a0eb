cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_gpu_dist.py tests/test_gpu_slab.py tests/test_gpu_parity.py tests/test_gpu_full_size.py tests/test_gpu_resident.py -m gpu -q -x > gpurun_out/r06_t44.log 2>&1; grep -E "passed|failed" gpurun_out/r06_t44.log | tail -1
timeout 600 python tools/probes/slab_time.py 512 4 8 0 beam > gpurun_out/r06_slab_time_beam512_w8.log 2>&1; grep '"replicated"' gpurun_out/r06_slab_time_beam512_w8.log | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); o=d['replicated']; print(d['rank'], o['rows_ms'], o['wall_ms'])"; tail -1 gpurun_out/r06_slab_time_beam512_w8.log | cut -c1-600
