#!/usr/bin/env python3
"""Print (calls, average us, total ms, name) from a rocprofv3 *kernel_stats.csv, optionally filtered by a substring.  (development tool)"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for r in rows:
    if pat in r["Name"]:
        print(f"{int(r['Calls']):6d} {float(r['AverageNs']) / 1e3:10.1f} us {float(r['TotalDurationNs']) / 1e6:9.3f} ms  {r['Name'][:150]}")
