"""Timeline digest of a rocprofv3 --kernel-trace database: per stream (= per layer of a job) the kernels in dispatch order
with start (ms since the first dispatch of the window), duration and the gap since the previous kernel of the same stream ended;
plus, per kernel name, mean duration / mean gap-before, and the chip-level concurrency (kernels in flight, sampled every 50 us).
Usage: rocpd_timeline.py results.db [t0_ms t1_ms] [--streams N]   (window relative to the first dispatch; default: the last 40 ms)"""
import sqlite3
import sys
from collections import defaultdict


def main():
    db = sqlite3.connect(sys.argv[1])
    args = [a for a in sys.argv[2:] if not a.startswith("--")]
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]
    ks = [t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
    cols = [r[1] for r in db.execute("pragma table_info(%s)" % kd)]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    scol = "stream_id" if "stream_id" in cols else qcol
    rows = list(db.execute("select d.start, d.end, substr(s.display_name, 1, 60), d.%s, d.grid_size_x / d.workgroup_size_x "
                           "from %s d join %s s on d.kernel_id = s.id order by d.start" % (scol, kd, ks)))
    if not rows:
        print("no dispatches")
        return
    t_first, t_last = rows[0][0], max(r[1] for r in rows)
    anchor, jobs_back, njobs = None, 1, 12
    for a in sys.argv[2:]:
        if a.startswith("--anchor="):      # --anchor=k_lasso_prep:12 : the window is the last job = the last 12 dispatches of that kernel
            anchor = a.split("=", 1)[1]
            if ":" in anchor:
                anchor, njobs = anchor.split(":")[0], int(anchor.split(":")[1])
    if len(args) >= 2:
        w0, w1 = t_first + float(args[0]) * 1e6, t_first + float(args[1]) * 1e6
    elif anchor:
        hits = [r for r in rows if anchor in r[2]]
        w0 = hits[-njobs][0] - 50e3
        later = [r for r in rows if r[0] > w0 and "k_probe" not in r[2]]
        w1 = max(r[1] for r in later) + 50e3
    else:
        w0, w1 = t_last - 40e6, t_last
    win = [r for r in rows if r[0] >= w0 and r[0] < w1 and "k_probe" not in r[2]]
    print("window %.3f .. %.3f ms after the first dispatch, %d dispatches, %d streams" % (
        (w0 - t_first) / 1e6, (w1 - t_first) / 1e6, len(win), len({r[3] for r in win})))
    by_stream = defaultdict(list)
    for r in win:
        by_stream[r[3]].append(r)
    # per kernel name: duration and gap-before statistics
    stats = defaultdict(lambda: [0, 0.0, 0.0])
    for st, rs in by_stream.items():
        prev_end = None
        for (a, b, name, _, wg) in rs:
            s_ = stats[name]
            s_[0] += 1
            s_[1] += (b - a) / 1e3
            if prev_end is not None:
                s_[2] += max(0.0, (a - prev_end) / 1e3)
            prev_end = b
    print("| kernel | dispatches | mean duration us | mean gap since the previous kernel of its stream ended us |")
    print("|---|---|---|---|")
    for name, (n, d, g) in sorted(stats.items(), key=lambda kv: -kv[1][1]):
        print("| `%s` | %d | %.1f | %.1f |" % (name, n, d / n, g / n))
    # concurrency
    step = 50e3
    t = w0
    hist = defaultdict(int)
    ev = sorted([(r[0], 1) for r in win] + [(r[1], -1) for r in win])
    i, cur = 0, 0
    while t < w1:
        while i < len(ev) and ev[i][0] <= t:
            cur += ev[i][1]
            i += 1
        hist[cur] += 1
        t += step
    tot = sum(hist.values())
    print("kernels in flight (share of 50 us samples): " + ", ".join("%d: %.0f%%" % (k, 100.0 * v / tot) for k, v in sorted(hist.items())))
    # the busiest streams in full
    nshow = 2
    for a in sys.argv[2:]:
        if a.startswith("--streams"):
            nshow = int(a.split("=")[1]) if "=" in a else 2
    busiest = sorted(by_stream.items(), key=lambda kv: -sum(r[1] - r[0] for r in kv[1]))[:nshow]
    for st, rs in busiest:
        print("\nstream %s: %d dispatches" % (st, len(rs)))
        print("| start ms | dur us | gap us | wgs | kernel |")
        print("|---|---|---|---|---|")
        prev_end = None
        for (a, b, name, _, wg) in rs:
            gap = (a - prev_end) / 1e3 if prev_end is not None else 0.0
            print("| %.3f | %.1f | %.1f | %d | %s |" % ((a - w0) / 1e6, (b - a) / 1e3, gap, wg, name[:48]))
            prev_end = b


if __name__ == "__main__":
    main()
