"""Kernel timeline of the LAST call in a rocprofv3 --kernel-trace csv: start offset, duration, stream, gap to the kernel
that ended latest before it started.  usage: timeline.py <kernel_trace.csv> [first kernel name prefix of a call]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = sys.argv[2] if len(sys.argv) > 2 else "sk::k_make_keys"
starts = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith(first)]
lo = starts[-1] if starts else 0
t0 = int(rows[lo]["Start_Timestamp"])
busy_end = t0
idle = 0
for r in rows[lo:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = s - busy_end
    if gap > 0:
        idle += gap
    print("%9.1f us  +%7.1f us  gap %6.1f  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, max(gap, 0) / 1e3, r.get("Queue_Id", "?"), r["Kernel_Name"][:60]))
    busy_end = max(busy_end, e)
print("call span %.1f us, device idle inside it %.1f us" % ((busy_end - t0) / 1e3, idle / 1e3))
