"""Per-kernel matrix-pipe utilisation from one rocprofv3 --pmc database:

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
              SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 bench.py ...
    python tools/pmc_mfma.py results.db out.json

mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (SIMDs x elapsed shader cycles), elapsed cycles = GRBM_GUI_ACTIVE / 8 (the counter
is summed over the 8 XCDs), SIMDs = 1024.  The wave-cycle buckets (quad-cycles) are given as fractions of SQ_WAVE_CYCLES:
parked on s_waitcnt / barrier, stalled at issue, issuing.
"""
import json, sqlite3, sys

db = sqlite3.connect(sys.argv[1])
q = "select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection group by kernel_name, counter_name"
per = {}
for name, ctr, n, v, d in db.execute(q):
    k = per.setdefault(name, {"launches": n, "avg_ns_under_pmc": d})
    k[ctr] = v
rows = []
for name, k in per.items():
    gui = k.get("GRBM_GUI_ACTIVE")
    mf = k.get("SQ_VALU_MFMA_BUSY_CYCLES")
    wc = k.get("SQ_WAVE_CYCLES")
    if not gui or mf is None:
        continue
    cyc = gui / 8.0
    r = {"kernel": name, "launches": k["launches"], "avg_ns_under_pmc": k["avg_ns_under_pmc"], "elapsed_cycles": cyc,
         "clock_ghz": cyc / k["avg_ns_under_pmc"], "mfma_util": mf / (1024.0 * cyc)}
    if wc:
        for c, label in (("SQ_WAIT_ANY", "parked"), ("SQ_WAIT_INST_ANY", "issue_stalled"), ("SQ_ACTIVE_INST_ANY", "issuing"),
                         ("SQ_ACTIVE_INST_VALU", "issuing_valu")):
            if c in k:
                r[label + "_frac_of_wave_cycles"] = k[c] / wc
    rows.append(r)
rows.sort(key=lambda r: -r["avg_ns_under_pmc"] * r["launches"])
json.dump(rows, open(sys.argv[2], "w"), indent=1)
for r in rows[:14]:
    print(f"{r['launches']:5d} x {r['avg_ns_under_pmc']/1e3:8.1f} us  clock {r['clock_ghz']:.2f} GHz  MFMA util {100*r['mfma_util']:5.1f} %  "
          f"parked {100*r.get('parked_frac_of_wave_cycles', 0):4.1f} %  stalled {100*r.get('issue_stalled_frac_of_wave_cycles', 0):4.1f} %  "
          f"issuing {100*r.get('issuing_frac_of_wave_cycles', 0):4.1f} %  {r['kernel'][:70]}")
