"""Per-kernel summary (calls, mean / min / total us) from a rocprofv3 results .db."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); c = db.cursor()
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]; ks = [t for t in tabs if 'kernel_symbol' in t][0]
q = (f"select s.kernel_name, count(*), avg(d.end-d.start)/1e3, min(d.end-d.start)/1e3, sum(d.end-d.start)/1e3 "
     f"from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 5 desc")
print("kernel,calls,avg_us,min_us,total_us")
for r in c.execute(q): print(f"{r[0][:90]},{r[1]},{r[2]:.1f},{r[3]:.1f},{r[4]:.1f}")
