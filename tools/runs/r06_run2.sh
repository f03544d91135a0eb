# round 6, call 2: s_setprio placements in the brick kernel, wave balance of the row walk (beam 512^3, sheet 1024^3)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python tools/probes/prio_probe.py 512 > gpurun_out/r06_prio_probe.log 2>&1
SPMV_SCENE=sheet timeout 900 python tools/probes/prio_probe.py 1024 > gpurun_out/r06_prio_probe_sheet.log 2>&1
grep -v "rows differ" gpurun_out/r06_prio_probe.log | tail -24
