export AVS_DIST_TIMEOUT_MS=4000
AVS_CG_RESIDENT_TIMERS=200 timeout 300 python tools/loopback_scaling.py --worlds 8 --ranks 0,3 --iters 640 2>&1 | grep -a "200 iterations of work\|us_per_iter" | cut -c1-220
timeout 300 python tools/resident_probe.py beam128 hipbuckling 2>&1 | grep "tol 0.001" | cut -c90-200
