# Round-6 profiles of the FINAL binary (run on the GPU box through gpurun; the summaries are copied to profiles/ by hand):
#  1. rocprofv3 --kernel-trace --stats of the default bench (headline: k_spmv_brick inside the PCG loop)
#  2. separate --pmc passes (FETCH_SIZE | WRITE_SIZE | two SQ sets | TCP) for k_spmv_brick and k_update_r (the calibration kernel)
#  3. a counter record for profiles/spmv_counters.json carrying the fingerprint of this source tree: bench.py quotes it only while the
#     tree is unchanged (capi.source_fingerprint)
# MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE in separate passes; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B.
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/r06prof && mkdir -p $O
TAG=${1:-r06}
PREC=${2:-f64}     # f32: the float-vector loop (bench.py --precision f32): a second record with "f32": true is written
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o $TAG -- python $R/bench.py --precision $PREC --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $O/stats.log 2>&1; echo "stats rc=$?"
rm -f $O/stats/*/*kernel_trace.csv $O/stats/*kernel_trace.csv
f=$(ls $O/stats/*/${TAG}_kernel_stats.csv $O/stats/${TAG}_kernel_stats.csv 2>/dev/null | head -1); cp "$f" $O/${TAG}_kernel_stats.csv; head -6 $O/${TAG}_kernel_stats.csv | cut -c1-200
grep -h '"metric"' $O/stats.log > $O/${TAG}_bench_line_under_rocprof.json
pmc() { name=$1; re=$2; shift; shift; timeout 400 rocprofv3 --kernel-include-regex "$re" --pmc "$@" --output-format csv -d $O/pmc_$name -o p -- python $R/bench.py --precision $PREC --steps 1 --warmup 0 --max-iters 96 --no-cpu-baseline --no-extra > $O/pmc_$name.log 2>&1; echo "pmc $name rc=$?"; }
pmc fetch "spmv|update_r" FETCH_SIZE
pmc write "spmv|update_r" WRITE_SIZE
pmc sq_a "spmv_brick" SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS
pmc sq_b "spmv_brick" SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS
pmc tcp "spmv_brick" TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum
cd $R && python - "$TAG" "$PREC" <<'PY'
import csv, collections, glob, json, sys
sys.path.insert(0, '.')
from adaptiveviscositysolver_amd import capi
tag = sys.argv[1]
f32 = len(sys.argv) > 2 and sys.argv[2] == "f32"
out = {}
for f in sorted(glob.glob('gpurun_out/r06prof/pmc_*/**/p_counter_collection.csv', recursive=True)):
    name = f.split('/')[2]
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        k = (r['Kernel_Name'].split('(')[0][-90:], r['Counter_Name'])
        acc[k][0] += 1
        acc[k][1] += float(r['Counter_Value'])
    for k, v in sorted(acc.items()):
        out.setdefault(name, {}).setdefault(k[0], {})[k[1]] = {"dispatches": v[0], "mean": v[1] / v[0]}
json.dump(out, open(f'gpurun_out/r06prof/{tag}_pmc_summary.json', 'w'), indent=1)
def get(passname, kernel_sub, counter):
    for k, v in out.get(passname, {}).items():
        if kernel_sub in k and counter in v:
            return v[counter]["mean"]
    return None
us = None
for r in csv.DictReader(open(f'gpurun_out/r06prof/{tag}_kernel_stats.csv')):
    if 'k_spmv_brick<true' in r['Name'] and ('float' in r['Name']) == f32:
        us = float(r['AverageNs']) / 1e3
        calls = int(r['Calls'])
line = json.loads(open(f'gpurun_out/r06prof/{tag}_bench_line_under_rocprof.json').read().strip().splitlines()[-1])
n, nnz = line['config']['n_dofs'], line['config']['nnz']
fetch, write = get('pmc_fetch', 'spmv_brick<true', 'FETCH_SIZE'), get('pmc_write', 'spmv_brick<true', 'WRITE_SIZE')
cal_f, cal_w = get('pmc_fetch', 'update_r', 'FETCH_SIZE'), get('pmc_write', 'update_r', 'WRITE_SIZE')
valu, waves = get('pmc_sq_b', 'spmv_brick<true', 'SQ_INSTS_VALU'), get('pmc_sq_b', 'spmv_brick<true', 'SQ_WAVES')
wait, wcyc = get('pmc_sq_a', 'spmv_brick<true', 'SQ_WAIT_ANY'), get('pmc_sq_a', 'spmv_brick<true', 'SQ_WAVE_CYCLES')
ldsc, ldsa = get('pmc_sq_a', 'spmv_brick<true', 'SQ_LDS_BANK_CONFLICT'), get('pmc_sq_a', 'spmv_brick<true', 'SQ_LDS_IDX_ACTIVE')
rec = {"kernel": "k_spmv_brick<DOT=true> inside the PCG loop, " + line['roofline']['kernel'][:200],
       "n": n, "nnz": nnz, "brick": True, "bytes_per_nonzero": 4, "tile_local_tables": False, "f32": f32,
       "source_sha16": capi.source_fingerprint(),
       "FETCH_SIZE_KB_mean": fetch, "WRITE_SIZE_KB_mean": write,
       "calibration_k_update_r": {"FETCH_SIZE_KB_mean": cal_f, "WRITE_SIZE_KB_mean": cal_w, "expected_bytes": (14 if f32 else 26) * n,
                                  "reported_bytes_with_x2_fetch": (2 * cal_f + cal_w) * 1024 if cal_f and cal_w else None},
       "correction": "MI355X_MICROARCH.md HBM section: on gfx950 FETCH_SIZE tallies 128-B requests at 64 B => x2 for coalesced streams (checked on k_update_r of the same run); units are KiB",
       "hbm_bytes_per_launch": (2 * fetch + write) * 1024 if fetch and write else None,
       "mean_kernel_us_rocprof": us, "kernel_calls": calls,
       "valu_wave_instructions_per_launch": valu,
       "valu_issue_frac": (valu * 4 / 1024 / (us * 1e-6 * 2.4e9)) if valu and us else None,
       "binding_resource": "valu_issue",
       "wait_frac": (wait / wcyc) if wait and wcyc else None,
       "lds_bank_conflict_frac_of_lds_active": (ldsc / ldsa) if ldsc and ldsa else None,
       "source": f"tools/profile_r06.sh {tag}: separate rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ sets), 96 dispatches each, of bench.py on this source tree; kernel time from profiles/{tag}_kernel_stats.csv; VALU issue = SQ_INSTS_VALU x 4 cycles / 1024 SIMDs over the kernel time at 2.4 GHz"}
json.dump([rec], open(f'gpurun_out/r06prof/{tag}_spmv_counters.json', 'w'), indent=1)
print(json.dumps(rec, indent=1)[:2500])
PY
rm -rf $O/pmc_*/ $O/stats/
