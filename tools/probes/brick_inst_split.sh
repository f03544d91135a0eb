# Round 6: instruction counts of k_spmv_brick<DOT> by phase (probe build, AVS_BRICK_DEBUG phase switches: 1 no halo fill, 2 no pattern rows,
# 4 no streamed rows), SQ counters per launch, 512^3 beam: where the VALU / SALU / LDS / VMEM instructions of a launch are issued.
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/r06_split && mkdir -p $O
for dbg in 0 1 2 3 7; do
  AVS_BRICK_DEBUG=$dbg SPMV_REPEATS=3 timeout 300 rocprofv3 --kernel-include-regex "spmv_brick<true" --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $O/d$dbg -o p -- python $R/tools/probes/spmv_time.py 512 > $O/d$dbg.log 2>&1
  echo "dbg $dbg rc=$?"
done
cd $R && python - <<'PY'
import csv, glob, collections
for dbg in (0, 1, 2, 3, 7):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(f'gpurun_out/r06_split/d{dbg}/**/p_counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r['Counter_Name']][0] += 1
            acc[r['Counter_Name']][1] += float(r['Counter_Value'])
    print("AVS_BRICK_DEBUG", dbg, {k: round(v[1] / v[0] / 1e6, 2) for k, v in sorted(acc.items())}, "(millions per launch)")
PY
rm -rf $O/d*/
