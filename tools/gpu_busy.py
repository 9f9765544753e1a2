"""GPU-busy fraction and inter-kernel gaps from a rocprofv3 --kernel-trace database, over the last `frac` of the dispatches
(the steady state of a bench run):    python tools/gpu_busy.py results.db out.json [frac]
busy = union of kernel intervals / span; gaps = idle intervals between consecutive kernels (overlapping kernels have none)."""
import json, sqlite3, sys

db = sqlite3.connect(sys.argv[1])
frac = float(sys.argv[3]) if len(sys.argv) > 3 else 0.4
rows = db.execute("select start, end, name from kernels order by start").fetchall()
rows = rows[int(len(rows) * (1 - frac)):]
span = max(r[1] for r in rows) - rows[0][0]
busy, cur_end, gaps = 0, rows[0][0], []
for s, e, _ in rows:
    if s > cur_end:
        gaps.append(s - cur_end)
        busy += e - s
        cur_end = e
    elif e > cur_end:
        busy += e - cur_end
        cur_end = e
gaps.sort()
out = {"kernels": len(rows), "span_ms": span / 1e6, "busy_fraction": busy / span, "sum_of_kernel_durations_ms": sum(e - s for s, e, _ in rows) / 1e6,
       "gaps": len(gaps), "gap_total_ms": sum(gaps) / 1e6, "gap_median_us": gaps[len(gaps) // 2] / 1e3 if gaps else 0.0,
       "gap_p90_us": gaps[int(len(gaps) * 0.9)] / 1e3 if gaps else 0.0, "gap_max_us": gaps[-1] / 1e3 if gaps else 0.0,
       "avg_kernel_us": sum(e - s for s, e, _ in rows) / len(rows) / 1e3}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(out)
