# where the value-indexed SpMV spends its time: wave stall split, LDS conflicts, texture-addresser and L1 stalls
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT
run() { name=$1; shift; timeout 120 rocprofv3 --kernel-include-regex "spmv_vi2" --pmc "$@" --output-format csv -d $R/gpurun_out/vi_$name -o p -- python $R/tools/gpu_probe.py --n 512 --levels 4 --variants 24 --repeats 3 --tol 1e-1 > $R/gpurun_out/vi_$name.log 2>&1; echo "pass $name rc=$?"; }
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS
run b SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS
run c TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
run d TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum
run e GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run f TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
cd $R && python - <<'PY'
import csv, collections, glob
for f in sorted(glob.glob('gpurun_out/vi_*/p_counter_collection.csv')):
    acc = collections.defaultdict(lambda:[0,0.0])
    for r in csv.DictReader(open(f)):
        k = (r['Kernel_Name'].split('(')[0][-56:], r['Counter_Name'])
        acc[k][0]+=1; acc[k][1]+=float(r['Counter_Value'])
    for k,v in sorted(acc.items()):
        if k[0].find("true, true, true") >= 0 or True:
            print("  %-58s %-36s n=%d mean=%.5g" % (k[0], k[1], v[0], v[1]/v[0]))
PY
