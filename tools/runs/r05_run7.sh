# value-code variant of the brick form (variable viscosity): correctness, then the 512^3 mu(x) beam with and without it
cd $GRAFT_REPO_ROOT && O=gpurun_out/r05g && mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_brick.py -x -q -m gpu > $O/brick.log 2>&1; echo "brick rc=$?"; tail -15 $O/brick.log
one() { python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra "$@" 2>$O/err.log | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']; print('it/s %.0f  ms/step %.1f  SpMV us %.1f  iterations %d stored MB %.0f brick %s asm ms %.1f' % (d['value'], d['ms_per_step'], r.get('mean_launch_us', 0), d['config']['cg_iterations_per_step'], r['stored_bytes_per_launch'] / 1e6, r.get('brick_form'), d['assembly_ms']['wall']))"; tail -2 $O/err.log | grep -i error; }
for vc in 0 1; do echo "== 512^3 mu(x) AVS_BRICK_VALUE_CODES=$vc"; AVS_BRICK_VALUE_CODES=$vc one --variable-viscosity; done 2>&1 | tee $O/varvisc512.log
for vc in 0 1; do echo "== 512^3 mu(x) AVS_BRICK_VALUE_CODES=$vc AVS_BRICK_TIMING"; AVS_BRICK_TIMING=1 AVS_BRICK_VALUE_CODES=$vc python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra --variable-viscosity 2>&1 | grep "brick build" | tail -12; done 2>&1 | tee $O/varvisc512_build.log
