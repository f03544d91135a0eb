cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_prepass.py tests/test_gpu_slab.py tests/test_precision_f32.py tests/test_gpu_matrix_formats.py tests/test_gpu_dist.py -m gpu -q > gpurun_out/r06_t_final4.log 2>&1; tail -2 gpurun_out/r06_t_final4.log | cut -c1-200
