# PMC passes restricted to the SpMV kernels (counter collection serialises every profiled dispatch).
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT
run() { name=$1; shift; timeout 150 rocprofv3 --kernel-include-regex "spmv" --pmc "$@" --output-format csv -d $R/gpurun_out/pmc_$name -o p -- python $R/tools/gpu_probe.py --n 512 --levels 4 --variants 1,7,8 --repeats 3 > $R/gpurun_out/pmc_$name.log 2>&1; echo "pass $name rc=$?"; }
run a TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum
run b TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum
run c SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE
run d TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum
cd $R && python - <<'PY'
import csv, collections, glob
for f in sorted(glob.glob('gpurun_out/pmc_*/p_counter_collection.csv')):
    acc = collections.defaultdict(lambda:[0,0.0])
    for r in csv.DictReader(open(f)):
        kn = r['Kernel_Name']
        if 'spmv' not in kn: continue
        k = (kn.split('(')[0][-60:], r['Counter_Name'])
        acc[k][0]+=1; acc[k][1]+=float(r['Counter_Value'])
    print(f)
    for k,v in sorted(acc.items()):
        print("  %-62s %-36s n=%d mean=%.4g" % (k[0], k[1], v[0], v[1]/v[0]))
PY
du -sh gpurun_out
