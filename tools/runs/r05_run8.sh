cd $GRAFT_REPO_ROOT && O=gpurun_out/r05g && mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_brick.py -x -q -m gpu -k "varvisc or sphere" > $O/brick_vc.log 2>&1; echo "brick vc rc=$?"; tail -30 $O/brick_vc.log
