"""Timeline of the last call in a rocprofv3 kernel trace csv: start offset, duration, gap to the previous kernel's end.
usage: timeline.py <kernel_trace.csv> [first kernel name substring = k_make_keys|k_gen]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = sys.argv[2] if len(sys.argv) > 2 else "k_check_sorted"
idx = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
lo = idx[-1]
t0 = int(rows[lo]["Start_Timestamp"])
prev_end = t0
for r in rows[lo:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f us  dur %7.1f  gap %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r["Kernel_Name"][:60]))
    prev_end = max(prev_end, e)
print("total %.1f us" % ((prev_end - t0) / 1e3))
