# planned walk (fitted cost model) against the strided walks INSIDE the PCG loop: headline beam, 512^3 sheet, 1024^3 sheet
cd $GRAFT_REPO_ROOT && O=gpurun_out/r05f && mkdir -p $O
one() { python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('it/s %.0f  ms/step %.1f  SpMV us %.1f  iterations %d walk %s' % (d['value'], d['ms_per_step'], d['roofline'].get('mean_launch_us', 0), d['config']['cg_iterations_per_step'], (d['roofline'].get('brick_form') or {}).get('walk')))"; }
for rep in 1 2; do for plan in 0 1; do
  echo "== headline AVS_BRICK_PLAN=$plan"; AVS_BRICK_PLAN=$plan one
done; done 2>&1 | tee $O/loop_plan_beam.log
for plan in 0 1; do echo "== sheet 512 AVS_BRICK_PLAN=$plan"; AVS_BRICK_PLAN=$plan one --config 5 --n 512 --levels 4; done 2>&1 | tee $O/loop_plan_sheet512.log
for plan in 0 1; do echo "== sheet 1024 AVS_BRICK_PLAN=$plan"; AVS_BRICK_PLAN=$plan one --config 5; done 2>&1 | tee $O/loop_plan_sheet1024.log
