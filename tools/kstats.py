#!/usr/bin/env python3
"""Print a rocprofv3 *_kernel_stats.csv compactly: calls, average us, name."""
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Name"].replace("void mfft::", "").replace("mfft::", "")
    print(f'{int(r["Calls"]):6d} {float(r["AverageNs"])/1e3:10.1f} us  {name[:110]}')
