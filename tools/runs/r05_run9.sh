cd $GRAFT_REPO_ROOT
for d in 1 2 4; do echo "== AVS_BRICK_DEBUG=$d"; AVS_BRICK_DEBUG=$d timeout 120 python tools/probes/vc_debug.py 0 2>&1 | grep -E "tiles|spmv|ERR|fault|Abort" | head -5; done
