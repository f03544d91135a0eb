cd $GRAFT_REPO_ROOT
AVS_TRACE_PHASES=1 timeout 300 python bench.py --scene beam --no-cpu-baseline --no-extra --steps 1 --warmup 1 2>&1 | grep "avs " | tail -14
