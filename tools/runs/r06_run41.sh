cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_prepass.py tests/test_scene_equivalents.py tests/test_precision_f32.py tests/test_gpu_full_size.py -m gpu -q -x > gpurun_out/r06_t41.log 2>&1; tail -3 gpurun_out/r06_t41.log | cut -c1-300
AVS_TRACE_PHASES=1 timeout 300 python bench.py --scene beam --no-cpu-baseline --no-extra --steps 1 --warmup 1 2>&1 | grep "avs rows\|avs assemble" | tail -4
AVS_TRACE_PHASES=1 timeout 300 python bench.py --no-cpu-baseline --no-extra --steps 1 --warmup 1 2>&1 | grep "avs rows\|avs assemble" | tail -4
