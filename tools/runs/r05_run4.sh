# planned (cost-balanced) walk of the brick kernel: calibration stamps of the strided walk, then planned vs strided on three scenes
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/r05d && mkdir -p $O
T=$R/tools/probes/spmv_time.py
{
for scene in beam sheet tank; do
  for plan in 0 1; do
    echo "== $scene AVS_BRICK_PLAN=$plan"; SPMV_SCENE=$scene AVS_BRICK_PLAN=$plan SPMV_REPEATS=100 timeout 200 python $T 512 2>&1 | grep -E "fused-dot|default SpMV|rror" | grep -v stream
    SPMV_SCENE=$scene AVS_BRICK_PLAN=$plan AVS_BRICK_DEBUG=80 AVS_BRICK_STAMP_FILE=$O/stamps_${scene}_plan$plan.bin SPMV_REPEATS=2 timeout 200 python $T 512 2>&1 | grep "brick phases" | tail -3
  done
done
} 2>&1 | tee $O/phases.log
python $R/tools/probes/brick_cost_fit.py $O/stamps_beam_plan0.bin $O/stamps_sheet_plan0.bin $O/stamps_tank_plan0.bin 2>&1 | tee $O/cost_fit.log
python $R/tools/probes/brick_cost_fit.py $O/stamps_beam_plan1.bin $O/stamps_sheet_plan1.bin $O/stamps_tank_plan1.bin 2>&1 | tee $O/cost_fit_plan1.log
