"""Kernel timeline of the tail of a rocprofv3 results .db: start offset, duration and gap to the previous kernel's end
for the last N dispatches (argv: db [N=200])."""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); c = db.cursor()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]; ks = [t for t in tabs if 'kernel_symbol' in t][0]
rows = list(c.execute(f"select s.kernel_name, d.start, d.end, d.grid_size_x, d.grid_size_y, d.grid_size_z, d.workgroup_size_x "
                      f"from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))[-n:]
t0, prev = rows[0][1], rows[0][1]
print("t_us,dur_us,gap_us,wgs,kernel")
for name, s, e, gx, gy, gz, wx in rows:
    short = re.sub(r"\.kd$", "", name)[:70]
    print(f"{(s - t0) / 1e3:9.1f},{(e - s) / 1e3:7.1f},{(s - prev) / 1e3:6.1f},{gx * gy * gz // max(wx, 1):6d},{short}")
    prev = e
