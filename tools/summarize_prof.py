"""Turns a rocprofv3 output tree under gpurun_out/ into the small tracked summaries under profiles/.

  python tools/summarize_prof.py gpurun_out/prof_r01 profiles/history/r01

Writes <out>_kernel_stats.csv (rocprofv3 --kernel-trace --stats), <out>_pmc.json (per-kernel FETCH_SIZE /
WRITE_SIZE per launch, with the gfx950 correction of MI355X_MICROARCH.md: FETCH_SIZE counts 64 B per 128 B
request on wide coalesced reads -> doubled; counters are in KiB).

  ... --pin  additionally writes profiles/pmc_current.json = {file, commit, sha256 of the conv kernel's sources}: the ONE summary
             bench.py's roofline.traffic may read, refused there as soon as one of those sources differs (run it on the tree the
             PMC passes were measured on, before editing further)."""
import collections
import csv
import glob
import json
import os
import sys

src, out = sys.argv[1], sys.argv[2]
PIN_SOURCES = ["lungmask_amd/csrc/nn_kernels_h3.hip", "lungmask_amd/csrc/nn_kernels.h", "lungmask_amd/csrc/nn_engine.hip", "lungmask_amd/csrc/lm_platform.h"]
stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
if stats:
    rows = list(csv.DictReader(open(stats[0])))
    with open(out + "_kernel_stats.csv", "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_ms", "avg_us", "pct"])
        for r in rows:
            w.writerow([r["Name"], r["Calls"], round(float(r["TotalDurationNs"]) / 1e6, 3), round(float(r["AverageNs"]) / 1e3, 2), r["Percentage"]])
pmc = {}
for kind, key in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    files = glob.glob(os.path.join(src, kind, "*", "*_counter_collection.csv"))
    if not files:
        continue
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(files[0])):
        if r["Counter_Name"] != key:
            continue
        a = agg[r["Kernel_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    for k, (n, v) in agg.items():
        pmc.setdefault(k, {})[key + "_KiB_per_launch"] = v / n
        pmc[k]["launches_" + key] = n
for k, d in pmc.items():
    f, w = d.get("FETCH_SIZE_KiB_per_launch"), d.get("WRITE_SIZE_KiB_per_launch")
    if f is not None and w is not None:
        d["hbm_bytes_per_launch_corrected"] = (2.0 * f + w) * 1024.0
wl = os.path.join(src, "pmc_workload.txt")
json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes; FETCH doubled per MI355X_MICROARCH.md (gfx950 counts 64 B per 128 B request)",
           "workload": open(wl).read().strip() if os.path.exists(wl) else "",
           "kernels": pmc}, open(out + "_pmc.json", "w"), indent=1)
print("wrote", out + "_kernel_stats.csv", out + "_pmc.json")

if "--pin" in sys.argv[3:] and pmc:
    import hashlib
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    commit = subprocess.run(["git", "rev-parse", "--short=12", "HEAD"], cwd=root, stdout=subprocess.PIPE, text=True).stdout.strip()
    dirty = subprocess.run(["git", "status", "--porcelain", "--"] + PIN_SOURCES, cwd=root, stdout=subprocess.PIPE, text=True).stdout.strip()
    pin = {"file": os.path.basename(out) + "_pmc.json", "commit": commit + ("+uncommitted kernel edits" if dirty else ""),
           "sources_sha256": {rel: hashlib.sha256(open(os.path.join(root, rel), "rb").read()).hexdigest() for rel in PIN_SOURCES}}
    json.dump(pin, open(os.path.join(root, "profiles", "pmc_current.json"), "w"), indent=1)
    print("pinned", pin["file"], "at", pin["commit"])
