#!/usr/bin/env python
"""Per-launch view of a rocprofv3 --kernel-trace database (rocpd SQLite):
every kernel of a time window in start order with its duration, the gap to
the previous kernel on the same stream/queue, and a per-name summary.

    python tools/trace_dump.py <results.db> [--last-ms 12] [--out file.txt]

Used to see where a training step's wall time goes (kernel time vs gaps)."""
import argparse
import sqlite3
import sys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--last-ms", type=float, default=0.0,
                    help="only the last N ms of the trace (0 = everything)")
    ap.add_argument("--skip-last-ms", type=float, default=0.0)
    ap.add_argument("--out", default=None)
    ap.add_argument("--max-rows", type=int, default=2500)
    args = ap.parse_args()
    db = sqlite3.connect(args.db)
    out = open(args.out, "w") if args.out else sys.stdout
    names = [r[0] for r in db.execute(
        "select name from sqlite_master where type in ('table','view')")]
    view = "kernels" if "kernels" in names else None
    if view is None:
        cand = [n for n in names if "kernel" in n.lower()]
        print("no `kernels` view; candidates:", cand, file=out)
        for n in cand[:6]:
            cols = [c[1] for c in db.execute("pragma table_info(%s)" % n)]
            print(n, cols, file=out)
        return
    cols = [c[1] for c in db.execute("pragma table_info(%s)" % view)]
    print("columns:", cols, file=out)

    def pick(*opts):
        for o in opts:
            if o in cols:
                return o
        return None
    c_name = pick("name", "kernel_name")
    c_start, c_end = pick("start", "start_time"), pick("end", "end_time")
    c_stream = pick("stream_id", "stream", "queue_id", "queue")
    sel = "select %s, %s, %s, %s from %s order by %s" % (
        c_name, c_start, c_end, c_stream or "0", view, c_start)
    rows = list(db.execute(sel))
    if not rows:
        print("empty trace", file=out)
        return
    t_end = max(r[2] for r in rows) - args.skip_last_ms * 1e6
    rows = [r for r in rows if r[2] <= t_end]
    if args.last_ms > 0:
        rows = [r for r in rows if r[1] >= t_end - args.last_ms * 1e6]
    t0 = rows[0][1]
    last_end = {}
    agg = {}
    busy = 0.0
    print("%10s %9s %9s %6s  %s" % ("t_us", "dur_us", "gap_us", "strm", "kernel"),
          file=out)
    for i, (name, s, e, st) in enumerate(rows):
        short = name.replace("(anonymous namespace)::", "").replace("pgnn::", "")
        short = short.replace("void ", "")
        depth, cut = 0, len(short)
        for j, ch in enumerate(short):
            if ch == "<":
                depth += 1
            elif ch == ">":
                depth -= 1
            elif ch == "(" and depth == 0:
                cut = j
                break
        short = short[:cut][:60]
        dur = (e - s) / 1e3
        gap = (s - last_end[st]) / 1e3 if st in last_end else 0.0
        last_end[st] = max(e, last_end.get(st, 0))
        a = agg.setdefault(short, [0, 0.0])
        a[0] += 1
        a[1] += dur
        busy += dur
        if i < args.max_rows:
            print("%10.1f %9.1f %9.1f %6s  %s" % ((s - t0) / 1e3, dur, gap,
                                                  st, short), file=out)
    span = (max(r[2] for r in rows) - t0) / 1e3
    print("\nwindow %.1f us, kernel time (summed over streams) %.1f us, "
          "%d launches" % (span, busy, len(rows)), file=out)
    print("\n%8s %10s %9s  %s" % ("calls", "total_us", "avg_us", "kernel"),
          file=out)
    for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("%8d %10.1f %9.1f  %s" % (n, t, t / n, k), file=out)


if __name__ == "__main__":
    main()
