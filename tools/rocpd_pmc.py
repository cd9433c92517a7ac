"""Per-kernel average of one PMC counter from a rocprofv3 rocpd database (counter run).
Usage: rocpd_pmc.py results.db COUNTER [kernel-substring]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    counter = sys.argv[2]
    filt = sys.argv[3] if len(sys.argv) > 3 else ""
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    pe = [t for t in tabs if t.startswith('rocpd_pmc_event')][0]
    ip = [t for t in tabs if t.startswith('rocpd_info_pmc')][0]
    kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
    ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
    scols = [r[1] for r in db.execute("pragma table_info(%s)" % ks)]
    name_col = "display_name" if "display_name" in scols else "kernel_name"
    q = ("select s.%s, count(*), avg(e.value), min(e.value), max(e.value) from %s e "
         "join %s p on e.pmc_id = p.id join %s d on e.event_id = d.event_id join %s s on d.kernel_id = s.id "
         "where p.name = ? group by s.%s order by 3 desc" % (name_col, pe, ip, kd, ks, name_col))
    import os
    print("commit: %s" % os.environ.get("CP_COMMIT", "unknown"))       # the library the pass profiled (bench.py: traffic_source)
    print("| kernel | dispatches | avg %s | min | max |" % counter)
    print("|---|---|---|---|---|")
    for name, n, avg, mn, mx in db.execute(q, (counter,)):
        if filt in name:
            print("| `%s` | %d | %.1f | %.1f | %.1f |" % (name[:100], n, avg, mn, mx))


if __name__ == "__main__":
    main()
