cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05a
timeout 900 python -m pytest tests/test_gpu_brick.py tests/test_gpu_cancel.py -x -q -m gpu > gpurun_out/r05a/brick_cancel.log 2>&1; echo "brick+cancel rc=$?"; tail -5 gpurun_out/r05a/brick_cancel.log
timeout 1200 python -m pytest tests/test_gpu_dist.py -x -q -m gpu > gpurun_out/r05a/dist.log 2>&1; echo "dist rc=$?"; tail -5 gpurun_out/r05a/dist.log
timeout 300 python bench.py --no-extra --no-cpu-baseline > gpurun_out/r05a/bench_headline.json 2> gpurun_out/r05a/bench_headline.err; echo "bench rc=$?"; cut -c1-600 gpurun_out/r05a/bench_headline.json
timeout 400 python bench.py --no-extra --no-cpu-baseline --force-dist > gpurun_out/r05a/bench_forcedist.json 2> gpurun_out/r05a/bench_forcedist.err; echo "force-dist rc=$?"; cut -c1-400 gpurun_out/r05a/bench_forcedist.json
