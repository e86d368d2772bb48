#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes of the bench command into profiles/traffic.json.

  rocprofv3 --pmc FETCH_SIZE --output-format csv -d D -o fetch -- python bench.py ...
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d D -o write -- python bench.py ...
  python tools/collect_traffic.py D <workload> <solver-kernel-substring>

FETCH_SIZE / WRITE_SIZE are in KiB.  Per MI355X_MICROARCH.md (section HBM) FETCH_SIZE
under-reports wide coalesced reads by exactly 2x on gfx950 and is otherwise uncalibrated;
the doubled figure is recorded as the estimate and the raw counters are kept next to it."""
import csv
import json
import os
import sys

d, workload, needle = sys.argv[1], sys.argv[2], sys.argv[3]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def mean_counter(path, name):
    vals = []
    with open(path) as fh:
        for row in csv.DictReader(fh):
            if needle in row["Kernel_Name"] and row["Counter_Name"] == name:
                vals.append(float(row["Counter_Value"]))
    return sum(vals) / len(vals), len(vals)


fetch, nf = mean_counter(os.path.join(d, "fetch_counter_collection.csv"), "FETCH_SIZE")
write, nw = mean_counter(os.path.join(d, "write_counter_collection.csv"), "WRITE_SIZE")
out_path = os.path.join(root, "profiles", "traffic.json")
try:
    rec = json.load(open(out_path))
except (OSError, ValueError):
    rec = {}
rec[workload] = {"kernel": needle, "launches_sampled": [nf, nw], "FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write,
                 "hbm_bytes_per_launch": (2.0 * fetch + write) * 1024.0,
                 "note": "(2*FETCH_SIZE + WRITE_SIZE) * 1024; separate --pmc passes"}
json.dump(rec, open(out_path, "w"), indent=1)
print(json.dumps(rec[workload]))
