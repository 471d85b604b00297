"""Summarise rocprofv3 --pmc CSV output per kernel (mean per dispatch)."""
import csv, glob, sys, collections
d = sys.argv[1]
for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].split("(")[0][:60]
        acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
    print("==", f)
    for k, v in acc.items():
        if not any(x in k for x in ("k_interp", "k_spread", "fft_rtc", "k_final", "k_advance")):
            continue
        print(k, {c: round(sum(x) / len(x), 1) for c, x in v.items()}, "n=%d" % len(next(iter(v.values()))))
