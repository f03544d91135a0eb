# k_spmv_brick with phases switched off (AVS_BRICK_DEBUG: 1 no fill, 2 no pattern rows, 4 no streamed rows, 16 phase stamps) and grid sizes
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/brick3
run() { tag=$1; shift; env "$@" timeout 300 python $R/tools/brick_probe.py --n ${N:-512} --levels 4 --repeats 30 > $R/gpurun_out/brick3/$tag.log 2>&1
  echo "$tag: $(grep -E 'brick_us|default_kernel_us|differing' $R/gpurun_out/brick3/$tag.log | tr -d '\n')"; grep "brick phases" $R/gpurun_out/brick3/$tag.log; tail -3 $R/gpurun_out/brick3/$tag.log | grep -i error; }
N=128 run v5_n128 AVS_BRICK_DEBUG=0
N=256 run v5_n256 AVS_BRICK_DEBUG=0
run v5_full AVS_BRICK_DEBUG=0
run v5_stamps AVS_BRICK_DEBUG=16
run v5_nopat AVS_BRICK_DEBUG=2
grep -E "max_runs|max_pattern|\"tiles\"|g_tiles|\"total\"|streamed|regular|global_pat" $R/gpurun_out/brick3/v5_full.log
