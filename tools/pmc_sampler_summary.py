"""Per-kernel summary of tools/profile_pmc_sampler.sh's passes over the short config-4 sampler run: launches, time share, MFMA busy, HBM-side
bytes per launch (FETCH_SIZE x 2 as MI355X_MICROARCH.md 'HBM' prescribes for gfx950, + WRITE_SIZE; both reported in KiB).  Writes
<dir>/summary.json (bench.py's secondary.roofline reads profiles/r04_pmc_sampler/summary.json).
    python3 tools/pmc_sampler_summary.py <dir> <tag>"""
import collections
import csv
import json
import sys
from pathlib import Path

d, tag = Path(sys.argv[1]), sys.argv[2]


def short(n):
    return n.split("(")[0].replace("void ", "").strip()


def per_kernel(path, counters):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] in counters:
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if "Scratch_Size" in r:
            acc[short(r["Kernel_Name"])]["_scratch"] = [float(r["Scratch_Size"])]
            acc[short(r["Kernel_Name"])]["_vgpr"] = [float(r["VGPR_Count"]) + float(r.get("Accum_VGPR_Count", 0) or 0)]
    return acc


stats = {}
for r in csv.DictReader(open(d / "kernel_stats.csv")):
    stats[short(r["Name"])] = dict(calls=int(r["Calls"]), total_ms=float(r["TotalDurationNs"]) / 1e6, mean_us=float(r["AverageNs"]) / 1e3,
                                   pct=float(r["Percentage"]))
fetch = per_kernel(d / "fetch_counter_collection.csv", {"FETCH_SIZE"})
write = per_kernel(d / "write_counter_collection.csv", {"WRITE_SIZE"})
mfma = per_kernel(d / "mfma_counter_collection.csv", {"SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"})
out = {}
print(f"{'kernel':52s} calls   total ms    %   MFMA busy  fetched(x2) MB  written MB  scratch B/lane")
for name, s in sorted(stats.items(), key=lambda kv: -kv[1]["total_ms"])[:24]:
    row = dict(s)
    mf, gui = mfma.get(name, {}).get("SQ_VALU_MFMA_BUSY_CYCLES"), mfma.get(name, {}).get("GRBM_GUI_ACTIVE")
    if mf and gui:
        n = min(len(mf), len(gui))
        row["mfma_busy"] = round(sum(m / (g / 8 * 1024) for m, g in zip(mf[:n], gui[:n])) / n, 4)
    f = fetch.get(name, {}).get("FETCH_SIZE")
    w = write.get(name, {}).get("WRITE_SIZE")
    if f:
        row["fetch_bytes_x2_per_launch"] = round(2 * 1024 * sum(f) / len(f))
    if w:
        row["write_bytes_per_launch"] = round(1024 * sum(w) / len(w))
    sc = fetch.get(name, {}).get("_scratch")
    if sc:
        row["scratch_bytes_per_lane"] = sc[0]
        row["vgprs"] = fetch[name]["_vgpr"][0]
    out[name] = row
    print(f"{name[:52]:52s} {s['calls']:5d} {s['total_ms']:9.2f} {s['pct']:5.1f}  "
          f"{100 * row['mfma_busy']:8.1f} %" if "mfma_busy" in row else f"{name[:52]:52s} {s['calls']:5d} {s['total_ms']:9.2f} {s['pct']:5.1f}  {'-':>10s}",
          f"{row.get('fetch_bytes_x2_per_launch', 0) / 1e6:13.1f} {row.get('write_bytes_per_launch', 0) / 1e6:11.1f} {row.get('scratch_bytes_per_lane', 0):10.0f}")
json.dump({"source": tag, "run": "tools/sampler_short.py (B=16, L=8192, S=3, cond_scale 2, bf16)", "kernels": out}, open(d / "summary.json", "w"), indent=1)
