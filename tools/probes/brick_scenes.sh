R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/brick4
for sc in "sphere 256" "hipbeam 0" "hipbuckling 0" "sheet 512" "beam 256"; do set -- $sc
 timeout 600 python $R/tools/brick_probe.py --scene $1 --n $2 --levels 4 --repeats 30 > $R/gpurun_out/brick4/$1_$2.log 2>&1
 echo "$1 $2: $(grep -E 'brick_us_dot|default_kernel_us|differing|regular_rows|\"rows\"|values|global_patterns|bits_ok|Error|error' $R/gpurun_out/brick4/$1_$2.log | tr -d '\n' | cut -c1-400)"
done
