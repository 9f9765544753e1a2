"""Average counter values per kernel from a rocprofv3 --pmc database: python tools/pmc_raw.py results.db [kernel substring]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
sub = sys.argv[2] if len(sys.argv) > 2 else ""
q = "select kernel_name, counter_name, count(*), avg(value), avg(duration) from counters_collection group by kernel_name, counter_name"
for name, ctr, n, v, d in db.execute(q):
    if sub in name:
        print(f"{name[:60]:60s} {ctr:28s} n={n:4d} avg={v:16.1f} dur_us={d/1e3:9.1f}")
