# Round-6 counters of the CU-resident loop on this source tree (review r05 #7): separate rocprofv3 --pmc passes (two SQ sets, FETCH_SIZE,
# WRITE_SIZE) of ONE 96-iteration solve per workload -> gpurun_out/r06prof/resident_counters.json, keyed by capi.source_fingerprint();
# copied to profiles/resident_counters.json by hand.  bench.py quotes a record only while the tree is unchanged.
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/r06prof && mkdir -p $O
pmc() { name=$1; shift; timeout 300 rocprofv3 --kernel-include-regex "cg_resident" --pmc "$@" --output-format csv -d $O/rpmc_$name -o p -- python $R/bench.py --steps 1 --warmup 0 --max-iters 96 --no-cpu-baseline --no-extra $EXTRA > $O/rpmc_$name.log 2>&1; echo "pmc $name rc=$?"; }
for w in beam buckling c2; do
  case $w in beam) EXTRA="--scene beam";; buckling) EXTRA="--scene buckling";; c2) EXTRA="--n 128 --levels 3";; esac
  pmc ${w}_a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS
  pmc ${w}_b SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS
  pmc ${w}_fetch FETCH_SIZE
  pmc ${w}_write WRITE_SIZE
done
cd $R && python - <<'PY'
import csv, collections, glob, json, sys
sys.path.insert(0, '.')
from adaptiveviscositysolver_amd import capi
recs = []
for w in ("beam", "buckling", "c2"):
    c = {}
    kernel = None
    for f in sorted(glob.glob(f'gpurun_out/r06prof/rpmc_{w}_*/**/p_counter_collection.csv', recursive=True)):
        acc = collections.defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(f)):
            kernel = r['Kernel_Name'].split('(')[0][-60:]
            acc[r['Counter_Name']][0] += 1
            acc[r['Counter_Name']][1] += float(r['Counter_Value'])
        for k, v in acc.items():
            c[k] = v[1] / v[0]
    line = None
    for f in glob.glob(f'gpurun_out/r06prof/rpmc_{w}_a.log'):
        for l in open(f):
            if l.startswith('{"metric"'):
                line = json.loads(l)
    if not c or not line or 'SQ_WAVES' not in c:
        print(w, "incomplete", sorted(c)); continue
    it = 96
    waves = c['SQ_WAVES']
    n, nnz = line['config']['n_dofs'], line['config']['nnz']
    fetch, write = c.get('FETCH_SIZE'), c.get('WRITE_SIZE')
    rec = {"workload": line['config']['workload'], "kernel": kernel, "n": n, "nnz": nnz, "source_sha16": capi.source_fingerprint(),
           "pmc_run": "bench.py <workload> --steps 1 --warmup 0 --max-iters 96: ONE dispatch = one 96-iteration solve (plus the one-time load of the matrix words into registers and the vector write-back)",
           "iterations_in_dispatch": it, "counters_per_dispatch": c,
           "per_wave_per_iteration": {"valu_instructions": c['SQ_INSTS_VALU'] / waves / it, "salu_instructions": c['SQ_INSTS_SALU'] / waves / it,
                                      "lds_instructions": c['SQ_INSTS_LDS'] / waves / it, "vmem_read_instructions": c['SQ_INSTS_VMEM_RD'] / waves / it,
                                      "vmem_write_instructions": c['SQ_INSTS_VMEM_WR'] / waves / it},
           "wave_cycles_per_iteration": c['SQ_WAVE_CYCLES'] * 4 / waves / it,
           "valu_issue_frac": c['SQ_INSTS_VALU'] * 4 / (c['SQ_WAVE_CYCLES'] * 4),   # VALU issue cycles of a wave over its cycles, x 4 waves per SIMD below
           "wait_frac": c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES'], "wait_inst_frac": c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES'],
           "lds_bank_conflict_frac_of_lds_active": c['SQ_LDS_BANK_CONFLICT'] / c['SQ_LDS_IDX_ACTIVE'],
           "hbm_bytes_per_iteration_upper": ((2 * fetch + write) * 1024 / it) if fetch and write else None,
           "source": "tools/profile_r06_resident.sh: separate rocprofv3 --pmc passes of bench.py on this source tree (SQ sets, FETCH_SIZE, WRITE_SIZE; FETCH x 2 per MI355X_MICROARCH.md)"}
    rec["valu_issue_frac"] = rec["valu_issue_frac"] * 4   # four waves share a SIMD
    recs.append(rec)
    print(w, n, nnz, "wait", round(rec["wait_frac"], 3), "valu/wave/it", round(rec["per_wave_per_iteration"]["valu_instructions"], 1), "hbm/it", rec["hbm_bytes_per_iteration_upper"])
json.dump({"note": "CU-resident PCG loop (csrc/avs_pcg_resident.inl), round-6 counters of this source tree", "workloads": recs}, open('gpurun_out/r06prof/resident_counters.json', 'w'), indent=1)
PY
rm -rf $O/rpmc_*/
