"""Reads a rocprofv3 kernel-trace CSV and reports GPU busy time (union of kernel intervals), summed kernel time and the wall span
over the last `frac` of the trace (skips warm-up). Tuning aid for the stream-pipelined mode."""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")))
rows.sort()
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
t_lo = rows[0][0] + (rows[-1][1] - rows[0][0]) * (1 - frac)
rows = [r for r in rows if r[0] >= t_lo]
span = rows[-1][1] - rows[0][0]
busy = 0; cur_s, cur_e = rows[0][0], rows[0][1]
for s, e, _, _ in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = sum(e - s for s, e, _, _ in rows)
by = {}
for s, e, k, q in rows:
    k = k.split("(")[0][:40]; by[k] = by.get(k, 0) + (e - s)
print("kernels %d span %.3f ms busy %.3f ms (%.1f%%) summed %.3f ms (avg concurrency %.2f) queues %s" % (
    len(rows), span / 1e6, busy / 1e6, 100.0 * busy / span, tot / 1e6, tot / busy, sorted(set(r[3] for r in rows))))
for k, v in sorted(by.items(), key=lambda kv: -kv[1])[:8]:
    print("  %-42s %.3f ms" % (k, v / 1e6))
