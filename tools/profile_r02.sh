# Round-2 profiles (run on the GPU box through gpurun; summaries are copied to profiles/ by hand):
#  1. rocprofv3 --kernel-trace --stats of the default bench and of the variable-viscosity bench
#  2. separate --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ stall split | TCP) for the SpMV AND the assembly chain
#     (k_rows, k_unique_rows, k_merge_rows, k_initial_guess[_coarse], k_apply_regular, k_permute_rows, k_edge_stencils)
# MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE in separate passes; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B.
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/r02prof && mkdir -p $O
TAG=${1:-r02}
stats() { name=$1; shift; timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$name -o $TAG -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $O/stats_$name.log 2>&1; echo "stats $name rc=$?"; rm -f $O/stats_$name/*/*kernel_trace.csv $O/stats_$name/*kernel_trace.csv; }
stats uniform
stats varvisc --variable-viscosity
KRE="spmv|k_update|k_rows|k_unique_rows|k_merge_rows|k_initial_guess|k_apply_regular|k_permute_rows|k_edge_stencils|k_center_stencils|k_tlt|k_cwin|k_vi_"
pmc() { name=$1; scene=$2; shift 2; timeout 400 rocprofv3 --kernel-include-regex "$KRE" --pmc "$@" --output-format csv -d $O/pmc_${scene}_$name -o p -- python $R/bench.py --steps 1 --warmup 0 --max-iters 96 --no-cpu-baseline $( [ $scene = varvisc ] && echo --variable-viscosity ) > $O/pmc_${scene}_$name.log 2>&1; echo "pmc $scene $name rc=$?"; }
for scene in uniform varvisc; do
  pmc fetch $scene FETCH_SIZE
  pmc write $scene WRITE_SIZE
done
pmc sq uniform SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_LDS
pmc tcp uniform TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE
pmc sq varvisc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_LDS
cd $R && python - <<'PY'
import csv, collections, glob, json, os
out = {}
for f in sorted(glob.glob('gpurun_out/r02prof/pmc_*/**/p_counter_collection.csv', recursive=True)):
    scene = f.split('pmc_')[1].split('_')[0]
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        k = (r['Kernel_Name'].split('(')[0][-90:], r['Counter_Name'])
        acc[k][0] += 1
        acc[k][1] += float(r['Counter_Value'])
    for k, v in sorted(acc.items()):
        out.setdefault(scene, {}).setdefault(k[0], {})[k[1]] = {"dispatches": v[0], "mean": v[1] / v[0]}
json.dump(out, open('gpurun_out/r02prof/pmc_summary.json', 'w'), indent=1)
print(json.dumps({s: sorted(v) for s, v in out.items()}, indent=0)[:3000])
PY
grep -h '"metric"' $O/stats_*.log | cut -c1-400
