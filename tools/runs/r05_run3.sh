# where does k_spmv_brick spend its time?  phase stamps + phase switches for the previous commit's kernel and v7, then TA / TCP / SQ counters of v7
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/r05c && mkdir -p $O
for lib in libavs_probe_ref.so libavs_probe.so; do
  for d in 0 2 7; do
    echo "== $lib AVS_BRICK_DEBUG=$d"; AVS_PROBE_LIB_PATH=$R/adaptiveviscositysolver_amd/$lib AVS_BRICK_DEBUG=$d SPMV_REPEATS=100 timeout 200 python $R/tools/probes/spmv_time.py 512 2>&1 | grep -E "fused-dot|default SpMV|rror" | tail -4
  done
  echo "== $lib stamps"; AVS_PROBE_LIB_PATH=$R/adaptiveviscositysolver_amd/$lib AVS_BRICK_DEBUG=80 SPMV_REPEATS=2 timeout 200 python $R/tools/probes/spmv_time.py 512 2>&1 | grep "brick phases" | tail -3
done 2>&1 | tee $O/phases.log
pmc() { name=$1; shift; SPMV_REPEATS=3 timeout 300 rocprofv3 --kernel-include-regex "spmv_brick" --pmc "$@" --output-format csv -d $O/pmc_$name -o p -- python $R/tools/probes/spmv_time.py 512 > $O/pmc_$name.log 2>&1; echo "pmc $name rc=$?"; }
pmc ta TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
pmc tcp TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum
pmc sqb SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAVES
pmc sqa SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS
pmc grbm GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
cd $R && python - <<'PY'
import csv, collections, glob, json
out = {}
for f in sorted(glob.glob('gpurun_out/r05c/pmc_*/**/p_counter_collection.csv', recursive=True)):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        k = (r['Kernel_Name'].split('(')[0][-60:], r['Counter_Name'])
        acc[k][0] += 1; acc[k][1] += float(r['Counter_Value'])
    for k, v in sorted(acc.items()):
        out.setdefault(k[0], {})[k[1]] = {"dispatches": v[0], "mean": v[1] / v[0]}
json.dump(out, open('gpurun_out/r05c/pmc_summary.json', 'w'), indent=1)
for k, v in out.items():
    print(k, {n: round(x['mean']) for n, x in v.items()})
PY
rm -rf $O/pmc_*/
