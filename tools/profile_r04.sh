# Round-4 profiles (run on the GPU box through gpurun; summaries are copied to profiles/ by hand):
#  1. rocprofv3 --kernel-trace --stats of the default bench (headline: k_spmv_brick inside the PCG loop, the brick-form builder kernels)
#  2. separate --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ split | LDS) for k_spmv_brick and k_update_r (the calibration kernel)
#  3. stats + SQ counters of the CU-resident loop on the two scene-equivalents
# MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE in separate passes; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B.
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/r04prof && mkdir -p $O
TAG=${1:-r04}
stats() { name=$1; shift; timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$name -o $TAG -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra "$@" > $O/stats_$name.log 2>&1; echo "stats $name rc=$?"; rm -f $O/stats_$name/*/*kernel_trace.csv $O/stats_$name/*kernel_trace.csv; }
stats uniform
if [ "${2:-all}" = "all" ]; then
stats beam --scene beam
stats buckling --scene buckling
fi
# (AVS_BRICK=1: under counter collection every dispatch carries milliseconds of overhead, which the auto mode's timing of the two forms would measure)
pmc() { name=$1; re=$2; shift; shift; AVS_BRICK=1 timeout 400 rocprofv3 --kernel-include-regex "$re" --pmc "$@" --output-format csv -d $O/pmc_$name -o p -- python $R/bench.py --steps 1 --warmup 0 --max-iters 96 --no-cpu-baseline --no-extra $EXTRA > $O/pmc_$name.log 2>&1; echo "pmc $name rc=$?"; }
EXTRA=""
pmc fetch "spmv|k_update_r" FETCH_SIZE
pmc write "spmv|k_update_r" WRITE_SIZE
pmc sq_a "spmv_brick" SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS
pmc sq_b "spmv_brick" SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS
if [ "${2:-all}" = "all" ]; then
EXTRA="--scene buckling"
pmc res_a "cg_resident" SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS
pmc res_b "cg_resident" SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS
pmc res_fetch "cg_resident" FETCH_SIZE
pmc res_write "cg_resident" WRITE_SIZE
EXTRA="--scene beam"
pmc resb_a "cg_resident" SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS
pmc resb_b "cg_resident" SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS
pmc resb_fetch "cg_resident" FETCH_SIZE
pmc resb_write "cg_resident" WRITE_SIZE
fi
cd $R && python - <<'PY'
import csv, collections, glob, json, os
out = {}
for f in sorted(glob.glob('gpurun_out/r04prof/pmc_*/**/p_counter_collection.csv', recursive=True)):
    tag = f.split('/')[2]
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        k = (r['Kernel_Name'].split('(')[0][-90:], r['Counter_Name'])
        acc[k][0] += 1
        acc[k][1] += float(r['Counter_Value'])
    for k, v in sorted(acc.items()):
        out.setdefault(tag, {}).setdefault(k[0], {})[k[1]] = {"dispatches": v[0], "mean": v[1] / v[0]}
json.dump(out, open('gpurun_out/r04prof/pmc_summary.json', 'w'), indent=1)
print(json.dumps(out, indent=0)[:3000])
PY
grep -h '"metric"' $O/stats_*.log | cut -c1-260
for n in uniform beam buckling; do f=$(ls $O/stats_$n/*/${TAG}_kernel_stats.csv $O/stats_$n/${TAG}_kernel_stats.csv 2>/dev/null | head -1); echo "== $n"; head -8 "$f" | cut -c1-220; cp "$f" $O/${TAG}_${n}_kernel_stats.csv; done
grep -h '"metric"' $O/stats_*.log > $O/${TAG}_bench_lines.jsonl
