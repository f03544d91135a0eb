# round 6, call 12: fill rule on more scenes (value-code variant: sphere), cost-model stamps
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python tools/probes/fill_rule.py sphere256 sphere512 beam512_mu sheet512_8 sheet512_4 tank512 beam256_mu > gpurun_out/r06_fill_rule2.log 2>&1; grep "rows/tile" gpurun_out/r06_fill_rule2.log
for sc in beam tank sheet; do
n=512; [ $sc = sheet ] && n=1024
SPMV_SCENE=$sc AVS_BRICK_DEBUG=80 AVS_BRICK_STAMP_FILE=$GRAFT_REPO_ROOT/gpurun_out/r06_stamps_$sc.bin SPMV_REPEATS=2 timeout 600 python tools/probes/spmv_time.py $n 2>&1 | grep "brick phases" | tail -3
done
python tools/probes/brick_cost_fit.py gpurun_out/r06_stamps_beam.bin gpurun_out/r06_stamps_tank.bin gpurun_out/r06_stamps_sheet.bin > gpurun_out/r06_cost_fit.log 2>&1
for sc in beam tank sheet; do python tools/probes/brick_cost_fit.py gpurun_out/r06_stamps_$sc.bin >> gpurun_out/r06_cost_fit.log 2>&1; done
cat gpurun_out/r06_cost_fit.log
rm -f gpurun_out/r06_stamps_*.bin
