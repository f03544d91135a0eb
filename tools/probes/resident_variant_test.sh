# A/B of a resident-loop variant: 8-way loop-back (ranks 0 and 3) and two single-GPU scenes.  Extra environment for the variant
# comes from the caller (e.g. AVS_CG_RESIDENT_COHERENT_FILL=1 bash tools/probes/resident_variant_test.sh)
export AVS_DIST_TIMEOUT_MS=4000
AVS_CG_RESIDENT_VERBOSE=1 AVS_CG_RESIDENT_TIMERS=200 timeout 300 python tools/loopback_scaling.py --worlds 8 --ranks 0,3 --iters 640 2>&1 | grep -a "200 iterations of work\|us_per_iter\|iteration 20" | cut -c1-260
timeout 300 python tools/resident_probe.py beam128 hipbuckling 2>&1 | grep "tol 0.001" | cut -c90-200
