# where k_spmv_brick spends its time: phase switches (AVS_BRICK_DEBUG), then wave stall split / LDS / TA counters
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT
N=${1:-512}
mkdir -p $R/gpurun_out/brick
for dbg in 0 1 2 4 8 3 6 7 15; do
  AVS_BRICK_DEBUG=$dbg timeout 300 python $R/tools/brick_probe.py --n $N --levels 4 --repeats 30 > $R/gpurun_out/brick/dbg_$dbg.log 2>&1
  echo "debug=$dbg $(grep -h 'workgroups per CU' $R/gpurun_out/brick/dbg_$dbg.log | head -1) $(grep -E 'brick_us|default_kernel_us' $R/gpurun_out/brick/dbg_$dbg.log | tr -d '\n')"
done
run() { name=$1; shift; timeout 300 rocprofv3 --kernel-include-regex "spmv_brick" --pmc "$@" --output-format csv -d $R/gpurun_out/brick/pmc_$name -o p -- python $R/tools/brick_probe.py --n $N --levels 4 --repeats 3 > $R/gpurun_out/brick/pmc_$name.log 2>&1; echo "pass $name rc=$?"; }
run a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS
run b SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS
run c TA_TA_BUSY_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
run e GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
cd $R && python - <<'PY'
import csv, collections, glob
for f in sorted(glob.glob('gpurun_out/brick/pmc_*/p_counter_collection.csv')):
    acc = collections.defaultdict(lambda:[0,0.0])
    for r in csv.DictReader(open(f)):
        k = (r['Kernel_Name'].split('(')[0][-40:], r['Counter_Name'])
        acc[k][0]+=1; acc[k][1]+=float(r['Counter_Value'])
    for k,v in sorted(acc.items()):
        print("  %-42s %-36s n=%d mean=%.5g" % (k[0], k[1], v[0], v[1]/v[0]))
PY
