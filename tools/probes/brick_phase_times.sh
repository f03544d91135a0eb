# k_spmv_brick with phases switched off (AVS_BRICK_DEBUG: 1 no fill, 2 no pattern rows, 4 no streamed rows, 16 phase stamps) and grid sizes
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/brick3
run() { tag=$1; shift; env "$@" timeout 300 python $R/tools/brick_probe.py --n ${N:-512} --levels 4 --repeats 30 > $R/gpurun_out/brick3/$tag.log 2>&1
  echo "$tag: $(grep -E 'brick_us|default_kernel_us|differing' $R/gpurun_out/brick3/$tag.log | tr -d '\n')"; grep "brick phases" $R/gpurun_out/brick3/$tag.log; }
run np_full AVS_BRICK_PERSIST=0
run p2_full AVS_BRICK_PERSIST=2
run p2_stamps AVS_BRICK_PERSIST=2 AVS_BRICK_DEBUG=16
run p2_skeleton AVS_BRICK_PERSIST=2 AVS_BRICK_DEBUG=7
run p2_g1024 AVS_BRICK_PERSIST=2 AVS_BRICK_GRID=1024
