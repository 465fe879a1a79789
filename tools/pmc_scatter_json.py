#!/usr/bin/env python
"""FETCH_SIZE / WRITE_SIZE passes of the standalone scatter-max
(tools/pmc_scatter.sh) -> profiles/<name>.json, the sidecar bench.py copies
into `roofline.traffic` when its workload (E, C, K) matches.

    python tools/pmc_scatter_json.py gpurun_out profiles/r02_pmc_scatter_max.json

Unit handling per MI355X_MICROARCH.md (HBM / rocprofv3): FETCH_SIZE and
WRITE_SIZE are KiB; on gfx950 FETCH_SIZE reports half of the bytes of a wide
coalesced streaming read, so HBM bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024."""
import json
import os
import sqlite3
import sys


def avg_counter(db_path, counter):
    db = sqlite3.connect(db_path)
    row = db.execute(
        "select avg(value), count(*), min(kernel_name) from counters_collection "
        "where kernel_name like '%scatter_max_kernel%' and counter_name = ?",
        (counter,)).fetchone()
    return row


def short_kernel_name(name):
    """'void (anonymous namespace)::scatter_max_kernel<4, 2, 8, true>(float
    const*, ...)' -> 'scatter_max_kernel<4, 2, 8, true>'."""
    name = (name or "").replace("(anonymous namespace)::", "")
    name = name.replace("pgnn::", "")
    if name.startswith("void "):
        name = name[5:]
    depth = 0
    for i, ch in enumerate(name):      # cut at the argument list, not at '<..(..)..>'
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            return name[:i]
    return name


WRITE_NOTE = (
    "WRITE_SIZE exceeds the 4*K*C-byte output because the two range-boundary "
    "runs of every wave are flushed with float atomic-max (csrc/scatter_max.hip)"
    ": E / rows_per_wave waves x 2 runs x C atomics, each counted by the TCC "
    "as a partial-line write, on top of the plain stores of the complete "
    "segments; the lowest() fill is a separate memset kernel and not in this "
    "kernel's counters")


def main(src, out):
    vals = {}
    kernel = None
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        avg, n, kernel = avg_counter(
            os.path.join(src, "pmc2_scatter_%s" % c, "pmc_results.db"), c)
        vals[c] = avg
        vals[c + "_launches"] = n
    # workload printed by tools/kernel_bench.py scatter (one JSON line)
    wl = None
    with open(os.path.join(src, "pmc2_scatter_FETCH_SIZE.log")) as f:
        for line in f:
            if line.startswith("{"):
                wl = json.loads(line).get("workload")
    res = {
        "kernel": short_kernel_name(kernel),
        "workload": wl,
        "FETCH_SIZE_KiB": vals["FETCH_SIZE"],
        "WRITE_SIZE_KiB": vals["WRITE_SIZE"],
        "launches_averaged": vals["FETCH_SIZE_launches"],
        "write_note": WRITE_NOTE,
        "hbm_bytes_per_launch": (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024,
        "method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in "
                  "separate passes (tools/pmc_scatter.sh); bytes = (2*FETCH_SIZE "
                  "+ WRITE_SIZE)*1024: FETCH_SIZE counts half of a wide coalesced "
                  "read on gfx950 (MI355X_MICROARCH.md, HBM)",
    }
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
