#!/usr/bin/env python3
"""Print the top kernels of a rocprofv3 --stats run.  usage: kstats.py <dir>/<prefix>_kernel_stats.csv [rows=20]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 20]:
    print(f"{r['Name'][:100]:100s} {r['Calls']:>6s} {float(r['TotalDurationNs'])/1e6:9.2f}ms avg {float(r['AverageNs'])/1e3:8.1f}us {float(r['Percentage']):5.1f}%")
