#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd SQLite) kernel trace into CSV + markdown.

    python tools/prof_summary.py gpurun_out/prof/bench_results.db profiles/r01_bench_kernel_stats

rocprofv3 in ROCm 7.2 writes a rocpd database by default; its `top_kernels`
view is the `--stats` table (name, calls, total/avg duration in microseconds)."""
import csv
import sqlite3
import sys


def main(db_path, out_prefix, note=""):
    db = sqlite3.connect(db_path)
    rows = list(db.execute(
        "select name, total_calls, total_duration, average, percentage "
        "from top_kernels order by total_duration desc"))
    with open(out_prefix + ".csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
        for r in rows:
            w.writerow(r)
    with open(out_prefix + ".md", "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats summary\n\n")
        if note:
            f.write(note + "\n\n")
        f.write("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|\n")
        for name, calls, tot, avg, pct in rows:
            short = name.replace("(anonymous namespace)::", "").replace("pgnn::", "")
            short = short.split("(")[0][:70]
            f.write("| `%s` | %d | %.3f | %.2f | %.2f |\n" % (
                short, calls, tot / 1e3, avg, pct))
    print(out_prefix + ".md")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], " ".join(sys.argv[3:]))
