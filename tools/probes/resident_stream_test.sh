# the 4-way partition (streamed rows) in loop-back, ranks 0 and 2; extra environment from the caller
export AVS_DIST_TIMEOUT_MS=6000
AVS_CG_RESIDENT_VERBOSE=1 AVS_CG_RESIDENT_TIMERS=200 timeout 600 python tools/loopback_scaling.py --worlds 4 --ranks 0,2 --iters 640 2>&1 | grep -a "avs resident\] \(plan:\|streamed\|not\|iteration 20\|200 it\)\|us_per_iter" | cut -c1-300
