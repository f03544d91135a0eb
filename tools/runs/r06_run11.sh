# round 6, call 11: instruction split of the brick kernel by phase; where the brick form beats the word stream now (fill rule)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
bash tools/probes/brick_inst_split.sh > gpurun_out/r06_inst_split.log 2>&1; tail -8 gpurun_out/r06_inst_split.log
timeout 1500 python tools/probes/fill_rule.py > gpurun_out/r06_fill_rule.log 2>&1; grep "rows/tile" gpurun_out/r06_fill_rule.log
