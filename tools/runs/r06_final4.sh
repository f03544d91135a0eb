cd $GRAFT_REPO_ROOT
bash tools/runs/r06_final.sh 2>&1 | grep -E '"metric"' | cut -c1-400
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_prepass.py tests/test_gpu_slab.py tests/test_gpu_precision_f32.py tests/test_gpu_matrix_formats.py -q > gpurun_out/r06_t_final4.log 2>&1; tail -2 gpurun_out/r06_t_final4.log | cut -c1-200
