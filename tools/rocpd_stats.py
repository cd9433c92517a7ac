"""Summarise a rocprofv3 rocpd SQLite result (kernel trace) into a per-kernel table:
calls, total / avg / min / max duration (us), share.  Usage: rocpd_stats.py results.db [out.md]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in db.execute("pragma table_info(%s)" % kd)]
    scols = [r[1] for r in db.execute("pragma table_info(%s)" % ks)]
    name_col = "display_name" if "display_name" in scols else ("kernel_name" if "kernel_name" in scols else "name")
    q = ("select s.%s, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
         "from %s d join %s s on d.kernel_id = s.id group by s.%s order by 3 desc" % (name_col, kd, ks, name_col))
    rows = list(db.execute(q))
    total = sum(r[2] for r in rows) or 1
    lines = ["| kernel | calls | total us | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for name, n, tot, mn, mx in rows:
        short = name if len(name) < 110 else name[:107] + "..."
        lines.append("| `%s` | %d | %.1f | %.2f | %.2f | %.2f | %.1f |" % (
            short, n, tot / 1e3, tot / n / 1e3, mn / 1e3, mx / 1e3, 100.0 * tot / total))
    text = "\n".join(lines)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text + "\n")
    print(text)
    del cols


if __name__ == "__main__":
    main()
