#!/usr/bin/env python
"""One training step out of a rocprofv3 --kernel-trace database of
`bench.py --train`: steps are cut at `sgd_kernel`; per step the launches,
busy time and gaps of the step's stream, and (--out) the kernels of one step
of BOTH streams in start order.

    rocprofv3 --kernel-trace -d /tmp/p -o run -- python bench.py --train ...
    python tools/train_step_trace.py /tmp/p/.../run_results.db [--out file]
"""
import argparse
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--out", default=None)
    ap.add_argument("--step", type=int, default=-3,
                    help="which step to dump (index into the sgd launches)")
    args = ap.parse_args()
    db = sqlite3.connect(args.db)
    rows = list(db.execute(
        "select name, start, end, stream_id from kernels order by start"))
    sgd = [r for r in rows if "sgd_kernel" in r[0]]
    if len(sgd) < 4:
        print("fewer than 4 sgd_kernel launches in the trace")
        return
    st = sgd[0][3]
    print("%d steps; step stream id %d" % (len(sgd), st))
    print("%5s %9s %9s %9s %9s %8s" % ("step", "span_us", "busy_us", "gaps_us",
                                       "launches", "other"))
    for i in range(1, len(sgd)):
        lo, hi = sgd[i - 1][2], sgd[i][2]
        mine = [r for r in rows if r[3] == st and lo <= r[1] < hi]
        other = [r for r in rows if r[3] != st and lo <= r[1] < hi]
        busy = sum(r[2] - r[1] for r in mine) / 1e3
        gaps = sum(max(0, b[1] - a[2]) for a, b in zip(mine[:-1], mine[1:])) / 1e3
        print("%5d %9.1f %9.1f %9.1f %9d %8d" % (
            i, (hi - lo) / 1e3, busy, gaps, len(mine), len(other)))
    if args.out:
        i = args.step % len(sgd)
        lo, hi = sgd[i - 1][2], sgd[i][2]
        last = {}
        with open(args.out, "w") as f:
            f.write("one training step (sgd_kernel to sgd_kernel), both "
                    "streams, start order\n%10s %8s %8s %4s  kernel\n" % (
                        "t_us", "dur_us", "gap_us", "strm"))
            for n, s, e, sid in rows:
                if not (lo <= s < hi):
                    continue
                gap = (s - last[sid]) / 1e3 if sid in last else 0.0
                last[sid] = e
                short = n.replace("(anonymous namespace)::", "").replace(
                    "pgnn::", "").split("(")[0][:60]
                f.write("%10.1f %8.1f %8.1f %4d  %s\n" % (
                    (s - lo) / 1e3, (e - s) / 1e3, gap, sid, short))


if __name__ == "__main__":
    main()
