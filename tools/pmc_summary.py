#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel: launches and median value.
    python tools/pmc_summary.py FILE.csv [substring-filter]
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB."""
import csv
import statistics
import sys
from collections import defaultdict

csv.field_size_limit(1 << 30)
rows = defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].replace("void mfft::", "").replace("mfft::", "")[:90]
    rows[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for (name, counter), vals in sorted(rows.items()):
    if flt in name:
        print(f"{len(vals):5d}  {counter:12s} median {statistics.median(vals):14.1f}  min {min(vals):14.1f}  max {max(vals):14.1f}  {name}")
