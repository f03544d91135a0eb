cd $GRAFT_REPO_ROOT && O=gpurun_out/r05h && mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_dist.py -x -q -m gpu -k "brick" > $O/dist_brick.log 2>&1; echo "dist brick rc=$?"; grep -E "passed|failed|Error" $O/dist_brick.log | tail -5
timeout 600 python -m pytest tests/test_gpu_brick.py tests/test_gpu_matrix_formats.py -x -q -m gpu > $O/brick.log 2>&1; echo "brick rc=$?"; grep -E "passed|failed|Error" $O/brick.log | tail -5
