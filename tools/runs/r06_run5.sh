# round 6, call 5: the priorities once more, three alternations in one process
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocm-smi --showclocks --showpower 2>/dev/null | head -30 > gpurun_out/r06_smi_$1.log
timeout 900 python tools/probes/prio_probe.py 512 > gpurun_out/r06_prio_abc_$1.log 2>&1
grep -v "rows differ" gpurun_out/r06_prio_abc_$1.log | grep "us:" 
