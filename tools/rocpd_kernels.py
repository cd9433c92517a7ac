"""Per-kernel table from a rocprofv3 rocpd database: calls, avg/total duration, workgroups, threads/WG.
Usage: rocpd_kernels.py results.db [passes]  (passes: divide call counts / totals by it)"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
passes = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
q = ("select substr(s.display_name,1,64), count(*), avg(d.end-d.start)/1000.0, "
     "avg(d.grid_size_x*d.grid_size_y*d.grid_size_z/(d.workgroup_size_x*d.workgroup_size_y*d.workgroup_size_z)), "
     "d.workgroup_size_x, sum(d.end-d.start)/1000.0 from %s d join %s s on d.kernel_id=s.id "
     "group by s.display_name, d.workgroup_size_x order by 6 desc" % (kd, ks))
print("| kernel | calls/pass | avg us | workgroups | threads | total us/pass |")
print("|---|---|---|---|---|---|")
for name, n, avg, wgs, wsz, tot in db.execute(q):
    print("| `%s` | %.1f | %.1f | %.0f | %d | %.0f |" % (name, n / passes, avg, wgs, wsz, tot / passes))
