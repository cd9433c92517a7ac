"""Effective shader clock per kernel from a rocprofv3 counter pass: GRBM_GUI_ACTIVE (cycles the GPU was busy during the
dispatch) / the dispatch's duration.  Counter passes serialise the dispatches, so this is each kernel ALONE on the chip.
Usage: rocpd_clock.py results.db [kernel-substring]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    filt = sys.argv[2] if len(sys.argv) > 2 else ""
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    pe = [t for t in tabs if t.startswith('rocpd_pmc_event')][0]
    ip = [t for t in tabs if t.startswith('rocpd_info_pmc')][0]
    kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
    ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
    scols = [r[1] for r in db.execute("pragma table_info(%s)" % ks)]
    name_col = "display_name" if "display_name" in scols else "kernel_name"
    q = ("select s.%s, count(*), avg(e.value), avg(d.end - d.start), sum(e.value), sum(d.end - d.start) from %s e "
         "join %s p on e.pmc_id = p.id join %s d on e.event_id = d.event_id join %s s on d.kernel_id = s.id "
         "where p.name = 'GRBM_GUI_ACTIVE' group by s.%s order by 6 desc" % (name_col, pe, ip, kd, ks, name_col))
    print("| kernel | dispatches | avg GRBM_GUI_ACTIVE cycles | avg duration us | effective GHz (sum cycles / sum ns) |")
    print("|---|---|---|---|---|")
    for name, n, cyc, dur, scyc, sdur in db.execute(q):
        if filt in name and sdur:
            print("| `%s` | %d | %.0f | %.1f | %.3f |" % (name[:90], n, cyc, dur / 1e3, scyc / sdur))


if __name__ == "__main__":
    main()
