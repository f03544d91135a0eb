# Kernel-level profile of ONE rank of the W-way partition in loop-back mode (tools/loopback_scaling.py): where do the
# microseconds of an iteration go at 0.93 M rows per rank?  usage: bash tools/profile_loopback.sh [world] [rank] [tag]
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && W=${1:-8} && K=${2:-3} && TAG=${3:-r03} && O=$R/gpurun_out/${TAG}_loopprof && mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/w${W}r${K} -o lb -- python $R/tools/loopback_scaling.py --worlds $W --ranks $K --iters 640 > $O/w${W}r${K}.json 2> $O/w${W}r${K}.err
echo "rc=$?"
f=$(ls $O/w${W}r${K}/*/lb_kernel_stats.csv $O/w${W}r${K}/lb_kernel_stats.csv 2>/dev/null | head -1)
head -12 "$f" | cut -c1-220
rm -f $O/w${W}r${K}/*/*kernel_trace.csv $O/w${W}r${K}/*kernel_trace.csv
