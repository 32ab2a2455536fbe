"""Per-kernel sums of rocprofv3 --pmc counters: python tools/pmc_summary.py <dir> [kernel-substring]"""
import collections, csv, glob, sys
d = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:48]
        if sub not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (r["Dispatch_Id"]) not in seen: seen.add(r["Dispatch_Id"]); n[k] += 1
for k, c in agg.items():
    print(k, "dispatches", n[k])
    for name, v in sorted(c.items()): print(f"   {name:32s} {v:16.0f}")
