#!/usr/bin/env python3
"""Print the last N kernel records of a rocprofv3 kernel trace csv (start offset, duration, gap)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"]); prev = t0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  gap {(s - prev) / 1e3:6.1f}  grid {r['Grid_Size_X']:>8} wg {r['Workgroup_Size_X']:>4}  {r['Kernel_Name'][:70]}")
    prev = e
