"""Turns the raw rocprofv3 output of tools/profile_round.sh into small committed summaries."""
import collections, csv, glob, json, os, sys

out = sys.argv[1]


def pmc(dirname):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(out, dirname, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            name = row["Kernel_Name"].split("(")[0]
            acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} | {"dispatches": len(next(iter(d.values())))}
            for k, d in acc.items()}


stats = []
for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
    stats = list(csv.DictReader(open(f)))
summary = {"kernel_stats": [{"name": r["Name"].split("(")[0], "calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3,
                             "pct": float(r["Percentage"])} for r in stats],
           "pmc_fetch": pmc("pmc_fetch"), "pmc_write": pmc("pmc_write"), "pmc_sq": pmc("pmc_sq")}
# HBM traffic per launch of the hot kernels: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts
# 64 B per 128-B request for wide coalesced reads (MI355X_MICROARCH.md, HBM section) -> doubled.
traffic = {}
for k, v in summary["pmc_fetch"].items():
    w = summary["pmc_write"].get(k, {})
    if "FETCH_SIZE" in v:
        traffic[k] = {"fetch_bytes_raw": v["FETCH_SIZE"] * 1024, "fetch_bytes_corrected": 2 * v["FETCH_SIZE"] * 1024,
                      "write_bytes": w.get("WRITE_SIZE", 0.0) * 1024}
        traffic[k]["hbm_bytes_per_launch"] = traffic[k]["fetch_bytes_corrected"] + traffic[k]["write_bytes"]
summary["traffic"] = traffic
# which kernel sources these numbers belong to (bench.py only quotes them for the same sources)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
try:
    import bench
    summary["kernels_sha"] = bench.kernels_sha()
except Exception as e:  # noqa
    summary["kernels_sha"] = None
# dynamic instruction mix of the interpolation kernel per wave (vector issue is what bounds it)
f64 = pmc("pmc_f64")
summary["pmc_f64"] = f64
mix = {}
for k, v in summary["pmc_sq"].items():
    if "k_interp_b" in k and v.get("SQ_WAVES"):
        w = v["SQ_WAVES"]
        g = f64.get(k, {})
        wf = g.get("SQ_WAVES", w) or w
        mix = {"kernel": k, "valu_per_wave": v.get("SQ_INSTS_VALU", 0.0) / w,
               "fp64_fma_add_mul_per_wave": (g.get("SQ_INSTS_VALU_FMA_F64", 0.0) + g.get("SQ_INSTS_VALU_ADD_F64", 0.0) +
                                             g.get("SQ_INSTS_VALU_MUL_F64", 0.0)) / wf,
               "salu_per_wave": g.get("SQ_INSTS_SALU", 0.0) / wf, "smem_per_wave": g.get("SQ_INSTS_SMEM", 0.0) / wf,
               "lds_per_wave": g.get("SQ_INSTS_LDS", 0.0) / wf}
summary["interp_instmix"] = mix
json.dump(summary, open(os.path.join(out, "summary.json"), "w"), indent=1)
print("kernel                                   calls   avg_us    pct")
for r in summary["kernel_stats"][:14]:
    print("%-40s %6d %8.2f %6.2f" % (r["name"][:40], r["calls"], r["avg_us"], r["pct"]))
for k, t in traffic.items():
    if "sk::" in k:
        print("traffic %-30s fetch(corr) %.1f MB  write %.1f MB" % (k[:30], t["fetch_bytes_corrected"] / 1e6, t["write_bytes"] / 1e6))
try:
    print(open(os.path.join(out, "bench.json")).read())
except Exception as e:
    print("no bench.json", e)
