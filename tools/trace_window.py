"""Print a window of a rocprofv3 kernel trace (start, duration, gap to the previous kernel's end, name): development aid.
    python tools/trace_window.py kernel_trace.csv [n_dispatches]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
w = int(sys.argv[2]) if len(sys.argv) > 2 else 120
n = len(rows)
mid = rows[max(0, n // 2 - w // 2):n // 2 + w // 2]
t0 = int(mid[0]["Start_Timestamp"])
prev = None
for r in mid:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev else 0.0
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:6.1f}  gap {gap:6.1f}  {r['Kernel_Name'][:80]}")
    prev = e
