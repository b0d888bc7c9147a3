#!/usr/bin/env python3
"""Print the kernel sequence of the last bench step from a rocprofv3 kernel trace csv."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last step = from the last convert_f16/level-0 scan to the end
names = [r["Kernel_Name"] for r in rows]
starts = [i for i, n in enumerate(names) if "scan_filter_f32" in n or "scan_filter_i8" in n]
i0 = starts[-1] if starts else 0
while i0 > 0 and int(rows[i0]["Start_Timestamp"]) - int(rows[i0 - 1]["End_Timestamp"]) < 50000 and "select" not in names[i0 - 1]:
    i0 -= 1
t0 = int(rows[i0]["Start_Timestamp"])
prev_end = t0
for r in rows[i0:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  gap {(s - prev_end) / 1e3:6.1f}  grid {r['Grid_Size_X']:>8} wg {r['Workgroup_Size_X']:>4}  {r['Kernel_Name'][:70]}")
    prev_end = e
print(f"total {(prev_end - t0) / 1e3:.1f} us")
