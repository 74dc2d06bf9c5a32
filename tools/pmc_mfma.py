"""MFMA-busy share per kernel from a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE pass (csv output):
MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)   (MI355X_MICROARCH.md: the counter is in cycles,
32 per v_mfma_f32_32x32x16_bf16; GRBM_GUI_ACTIVE is summed over the 8 XCDs).  usage: python tools/pmc_mfma.py <counter_collection.csv>"""
import collections
import csv
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if any(k in name for k in ("mqa", "gemm_", "attn_delta")):
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
print(f"{'kernel':48s} launches  MFMA busy   GUI_ACTIVE/8 (cycles)")
for name, c in acc.items():
    mf, gui = c.get("SQ_VALU_MFMA_BUSY_CYCLES"), c.get("GRBM_GUI_ACTIVE")
    if not mf or not gui:
        continue
    n = min(len(mf), len(gui))
    util = sum(m / (g / 8 * 1024) for m, g in zip(mf[:n], gui[:n])) / n
    print(f"{name[:48]:48s} {n:8d}  {100 * util:8.1f} %  {sum(gui[:n]) / n / 8:14.0f}")
