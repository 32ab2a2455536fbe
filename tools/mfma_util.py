"""MFMA utilisation of the conv kernels from raw rocprofv3 counters (ROCm 7.2 has no gfx950 derived metrics):

  python tools/mfma_util.py gpurun_out/prof_r01i profiles/history/r01i

expects <src>/pmc_sq (SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY
SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE), <src>/pmc_grbm (GRBM_GUI_ACTIVE GRBM_COUNT) and
<src>/trace2 (--kernel-trace of the same command: durations without counters), each from its own pass
(tools/pmc_mfma.sh).  Counter values are sums over the 8 XCDs / all SIMDs of the device."""
import collections, csv, glob, json, os, sys

src, out = sys.argv[1], sys.argv[2]
N_CU, N_SIMD, N_XCD = 256, 4, 8


def counters(kind):
    f = glob.glob(os.path.join(src, kind, "*", "*_counter_collection.csv"))[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for r in csv.DictReader(open(f)):
        a = agg[r["Kernel_Name"]][r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    return {k: {c: v[1] / v[0] for c, v in d.items()} for k, d in agg.items()}, {k: next(iter(d.values()))[0] for k, d in agg.items()}


sq, launches = counters("pmc_sq")
gr, _ = counters("pmc_grbm")
dur = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(glob.glob(os.path.join(src, "trace2", "*", "*_kernel_trace.csv"))[0])):
    d = dur[r["Kernel_Name"]]
    d[0] += 1
    d[1] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
res = {"note": "per-launch averages over every launch of the kernel in `bench.py --streams 1 --steps 1 --warmup 1` (all layer shapes); "
               "counters are device-wide sums; mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (cycles per XCD x 256 CUs x 4 SIMDs); "
               "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_ANY count quad-cycles of resident waves",
       "kernels": {}}
for k, v in sq.items():
    if "conv_igemm" not in k:
        continue
    cyc_xcd = gr[k]["GRBM_GUI_ACTIVE"] / N_XCD
    us = dur[k][1] / dur[k][0] / 1e3
    busy = v["SQ_VALU_MFMA_BUSY_CYCLES"]
    res["kernels"][k] = {
        "launches": launches[k],
        "avg_duration_us": round(us, 1),
        "clock_GHz_under_profiler": round(cyc_xcd / us / 1e3, 3),
        "mfma_busy_frac": round(busy / (cyc_xcd * N_CU * N_SIMD), 4),
        "mfma_instructions_per_launch_32x32x16": round(busy / 32),
        "executed_PFLOPs": round(busy / 32 * 32768 / (us * 1e-6) / 1e15, 3),
        "wave_quad_cycles": round(v["SQ_WAVE_CYCLES"]),
        "frac_wave_issue_stalled_SQ_WAIT_INST_ANY": round(v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"], 3),
        "frac_wave_parked_SQ_WAIT_ANY": round(v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], 3),
        "frac_wave_issuing_SQ_ACTIVE_INST_ANY": round(v["SQ_ACTIVE_INST_ANY"] / v["SQ_WAVE_CYCLES"], 3),
        "lds_bank_conflict_frac_of_lds_cycles": round(v["SQ_LDS_BANK_CONFLICT"] / max(v["SQ_LDS_IDX_ACTIVE"], 1), 4),
        "raw_per_launch": {c: round(x) for c, x in v.items()},
        "GRBM_GUI_ACTIVE_per_launch": round(gr[k]["GRBM_GUI_ACTIVE"]),
    }
json.dump(res, open(out + "_mfma_util.json", "w"), indent=1)
for k, d in res["kernels"].items():
    print(k[:60], {a: d[a] for a in ("avg_duration_us", "clock_GHz_under_profiler", "mfma_busy_frac", "executed_PFLOPs", "frac_wave_parked_SQ_WAIT_ANY", "lds_bank_conflict_frac_of_lds_cycles")})
