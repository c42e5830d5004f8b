"""Prints the kernel timeline of the LAST proof in a rocprofv3 kernel-trace CSV of `bench.py --steps K` (single-proof mode):
start offset, duration, gap to the previous kernel's end, grid size. Tuning aid."""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("qp::", "")[:34],
                     int(r.get("Grid_Size", 0) or 0), int(r.get("Workgroup_Size", 0) or 0)))
rows.sort()
# a proof starts at the first k_salt_fill / first ntt kernel after a k_query_gather
ends = [i for i, r in enumerate(rows) if r[2].startswith("k_query_gather")]
lo = ends[-2] + 1 if len(ends) >= 2 else 0
hi = ends[-1] + 1
t0 = rows[lo][0]; prev_end = t0; tot_gap = 0; tot_dur = 0
for s, e, k, g, wg in rows[lo:hi]:
    gap = s - prev_end; tot_gap += max(gap, 0); tot_dur += e - s
    print("%9.1f us  dur %8.1f  gap %7.1f  %-34s grid %d" % ((s - t0) / 1e3, (e - s) / 1e3, gap / 1e3, k, g))
    prev_end = max(prev_end, e)
print("proof span %.1f us, kernel time %.1f us, gaps %.1f us, %d kernels" % ((rows[hi - 1][1] - t0) / 1e3, tot_dur / 1e3, tot_gap / 1e3, hi - lo))
