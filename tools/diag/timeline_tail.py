"""Kernel timeline of the tail of a rocprofv3 --kernel-trace csv: the last N launches (start offset, duration, gap to the
previous launch's end, name).  usage: timeline_tail.py <kernel_trace.csv> [N]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 110
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = t0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f us  +%6.1f us  gap %5.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, max(0, s - prev_end) / 1e3, r["Kernel_Name"].split("(")[0][:60]))
    prev_end = max(prev_end, e)
