cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 3300 python -m pytest tests -m gpu -q > gpurun_out/r06_t30_full.log 2>&1; tail -12 gpurun_out/r06_t30_full.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_smoke.log 2>&1; tail -3 gpurun_out/r06_smoke.log | cut -c1-300
