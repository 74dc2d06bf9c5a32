"""Per-step kernel summary from a rocprofv3 kernel trace of bench.py (tools/profile_bench.sh): wall / busy / idle-gap ms per step,
launches per step, per-kernel totals, and optionally per-grid medians of the kernels whose name contains a filter.
    python tools/trace_summary.py gpurun_out/prof_<tag> [--steps 4] [--grid substring]"""
import argparse, collections, csv, glob, sys
ap = argparse.ArgumentParser(); ap.add_argument("dir"); ap.add_argument("--steps", type=int, default=4); ap.add_argument("--grid", default=None)
ap.add_argument("--top", type=int, default=45); args = ap.parse_args()
f = glob.glob(args.dir + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
ad = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("adamw")]
seg = rows[ad[-args.steps - 1] + 1: ad[-1] + 1]                     # the last `steps` steps, optimizer to optimizer
K = args.steps
busy, cnt, grids = collections.Counter(), collections.Counter(), collections.defaultdict(list)
gaps, prev = 0, None
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"].split("(")[0][:58]
    busy[n] += e - s; cnt[n] += 1
    if prev is not None and s > prev: gaps += s - prev
    prev = max(prev or 0, e)
    if args.grid and args.grid in n:
        grids[(n, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]))].append((e - s) / 1e3)
wall = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / K / 1e6
print(f"wall {wall:.2f} ms/step  busy {sum(busy.values()) / K / 1e6:.2f}  idle gaps {gaps / K / 1e6:.2f}  launches/step {len(seg) / K:.0f}")
for n, v in busy.most_common(args.top):
    print(f"{n:58s} {cnt[n] // K:5d} {v / K / 1e6:8.3f}")
for k, v in sorted(grids.items()):
    v.sort(); print(k, "n/step", len(v) // K, "median us", round(v[len(v) // 2], 1), "ms/step", round(sum(v) / K / 1e3, 3))
