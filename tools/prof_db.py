"""Per-kernel calls and average duration (us) out of a rocprofv3 results database (gpurun_out/<dir>/*_results.db)."""
import glob
import sqlite3
import sys

for db in glob.glob(sys.argv[1] + "/*.db"):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    for r in c.execute(f"select s.kernel_name, count(*), avg(d.end-d.start)/1e3 from {kd} d join {ks} s on d.kernel_id=s.id group by 1 order by 3 desc"):
        print(f"{r[0][:90]:90s} {r[1]:5d} {r[2]:10.1f}")
