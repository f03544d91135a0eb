"""Timeline of the LAST assembly pass in a rocprofv3 kernel trace of bench.py: start offset, duration and the idle gap before
every kernel (host-side stalls show up as gaps).  usage: asm_timeline.py <dir with a_kernel_trace.csv>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/a_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if "k_dof_table" in r["Kernel_Name"] or "k_edge_stencils" in r["Kernel_Name"]]
# last assembly: from the last k_edge_stencils back to the first k_dof_table just before it (if any), to the first spmv after
last = max(i for i, r in enumerate(rows) if "k_edge_stencils" in r["Kernel_Name"])
i0 = last
while i0 > 0 and any(k in rows[i0 - 1]["Kernel_Name"] for k in ("k_dof_table", "k_check_table", "fillBuffer")):
    i0 -= 1
t0 = int(rows[i0]["Start_Timestamp"]); prev_end = t0; busy = 0
for r in rows[i0:]:
    name = r["Kernel_Name"]
    if "k_spmv" in name or "k_inv_diag" in name:
        break
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy += e - s
    print("%9.1f us  dur %8.1f  gap %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, name[:80]))
    prev_end = max(prev_end, e)
print("span %.2f ms, kernels %.2f ms" % ((prev_end - t0) / 1e6, busy / 1e6))
