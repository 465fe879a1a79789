#!/usr/bin/env python
"""Dump the kernel dispatches of a rocprofv3 rocpd database as CSV
(name, start_ns, end_ns, stream / queue ids) for timeline analysis off the box.

    python tools/timeline_dump.py run_results.db out.csv
"""
import csv
import sqlite3
import sys


def main(db_path, out_csv):
    db = sqlite3.connect(db_path)
    objs = list(db.execute(
        "select type, name from sqlite_master where type in ('table','view')"))
    cand = [n for t, n in objs if n == "kernels"] or \
           [n for t, n in objs if "kernel" in n.lower() and "top" not in n.lower()]
    print("objects:", [n for _, n in objs][:60], file=sys.stderr)
    for name in cand:
        cols = [r[1] for r in db.execute("pragma table_info(%s)" % name)]
        print(name, cols, file=sys.stderr)
        if "start" in cols and "end" in cols:
            keep = [c for c in ("name", "start", "end", "stream_id", "queue_id",
                                "stream", "queue", "grid_x", "workgroup_x",
                                "lds_size", "dispatch_id", "tid") if c in cols]
            rows = db.execute("select %s from %s order by start" % (
                ",".join(keep), name))
            with open(out_csv, "w", newline="") as f:
                w = csv.writer(f)
                w.writerow(keep)
                n = 0
                for r in rows:
                    r = list(r)
                    r[0] = str(r[0]).replace("(anonymous namespace)::", "") \
                        .replace("pgnn::", "").split("(")[0][:60]
                    w.writerow(r)
                    n += 1
            print("wrote", n, "rows to", out_csv, file=sys.stderr)
            return
    print("no kernel table with start/end found", file=sys.stderr)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
