"""Mean SQ counters per kernel from the two CSVs of tools/pmc_sq.sh:  python tools/sq_summary.py gpurun_out/<tag>"""
import collections
import csv
import sys
tot = collections.defaultdict(lambda: collections.defaultdict(list))
for f in ("a", "b"):
    for row in csv.DictReader(open(f"{sys.argv[1]}_sq_{f}.csv")):
        tot[row["Kernel_Name"][:70] + " grid " + row["Grid_Size"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
print("SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* count quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES and SQ_LDS_IDX_ACTIVE count cycles.")
for k, d in tot.items():
    print("==", k)
    m = {n: sum(v) / len(v) for n, v in d.items()}
    wc = m["SQ_WAVE_CYCLES"]
    for n, v in sorted(m.items()):
        print(f"  {n:28s} {v:16.0f}   / SQ_WAVE_CYCLES = {v / wc:6.3f}   (n={len(d[n])})")
