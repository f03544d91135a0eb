cd $GRAFT_REPO_ROOT
bash tools/runs/r06_final.sh 2>&1 | tail -12 | cut -c1-2700
cd $GRAFT_REPO_ROOT
for w in 8 4 2; do timeout 600 python tools/probes/slab_time.py 512 4 $w 0 beam > gpurun_out/r06_slab_time_beam512_w$w.log 2>&1; tail -1 gpurun_out/r06_slab_time_beam512_w$w.log | cut -c1-700; done
timeout 700 python tools/probes/slab_time.py 1024 5 8 0 sheet > gpurun_out/r06_slab_time_sheet1024_w8.log 2>&1; tail -1 gpurun_out/r06_slab_time_sheet1024_w8.log | cut -c1-700
timeout 3300 python -m pytest tests -m gpu -q > gpurun_out/r06_t_final_full.log 2>&1; tail -6 gpurun_out/r06_t_final_full.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_smoke.log 2>&1; tail -1 gpurun_out/r06_smoke.log | cut -c1-300
