# usage: BR="3 -1" VARS=14,16 bash tools/pmc_probe.sh  -- L2<->fabric read requests of SpMV variants
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT
for b in ${BR:-3}; do
  AVS_BRICK_SHIFT=$b timeout 200 rocprofv3 --kernel-include-regex "spmv" --pmc TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/pp_$b -o p -- python $R/tools/gpu_probe.py --n 512 --levels 4 --variants ${VARS:-14,16} --repeats 3 --tol 1e-1 > $R/gpurun_out/pp_$b.log 2>&1
  echo "brick shift $b rc=$?"; grep -E "spmv variant" $R/gpurun_out/pp_$b.log
  python - <<PY
import csv, collections
acc = collections.defaultdict(lambda:[0,0.0])
for r in csv.DictReader(open('$R/gpurun_out/pp_$b/p_counter_collection.csv')):
    k = (r['Kernel_Name'].split('(')[0][-58:], r['Counter_Name'])
    acc[k][0]+=1; acc[k][1]+=float(r['Counter_Value'])
ks = sorted(set(k[0] for k in acc))
for kn in ks:
    g = lambda c: acc[(kn,c)][1]/max(acc[(kn,c)][0],1)
    rd = g('TCC_EA0_RDREQ_128B_sum')*128 + g('TCC_EA0_RDREQ_64B_sum')*64
    print("  %-60s read %.1f MB  hit %.3g miss %.3g" % (kn, rd/1e6, g('TCC_HIT_sum'), g('TCC_MISS_sum')))
PY
done
