# rocprofv3 kernel-trace + stats of the default bench command (summary only is kept)
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final -o ${1:-r01d} -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_final.log 2>&1
cd $R && rm -f gpurun_out/prof_final/*kernel_trace.csv && grep -h '"metric"' gpurun_out/prof_final.log | cut -c1-1200
