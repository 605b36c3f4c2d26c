"""Fold a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE pass into per-kernel MFMA utilisation.
usage: mfma_busy_summary.py <pmc_dir> <out.json>
MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 256 CUs x 4 SIMDs): GRBM_GUI_ACTIVE comes back summed over the 8
XCDs (2.91 M per 175 us launch), busy cycles summed over all SIMDs (the gfx94x MfmaUtil formula otherwise; ROCm 7.2 ships no
gfx950 derived-counter section, MI355X_MICROARCH.md 'rocprofv3 PMC slots').  Busy cycles count 32 per 32x32x16 bf16 MFMA."""
import csv, glob, json, re, sys
from collections import defaultdict
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from pmc_summary import bench_name

acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(int)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "fo1::" not in k:
            continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            cnt[k] += 1
out = {"formula": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs)", "kernels": {}}
for k, c in acc.items():
    gui = c.get("GRBM_GUI_ACTIVE", 0.0)
    if gui <= 0 or c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) <= 0:
        continue
    name = bench_name(k)
    row = out["kernels"].setdefault(name, dict(launches=0, mfma_busy_cycles=0.0, gui_active_cycles=0.0))
    row["launches"] += cnt[k]
    row["mfma_busy_cycles"] += c["SQ_VALU_MFMA_BUSY_CYCLES"]
    row["gui_active_cycles"] += gui
for name, row in out["kernels"].items():
    row["mfma_util"] = round(row["mfma_busy_cycles"] / (row["gui_active_cycles"] / 8.0 * 1024.0), 4)
json.dump(out, open(sys.argv[2], "w"), indent=1, sort_keys=True)
for name, row in sorted(out["kernels"].items(), key=lambda kv: -kv[1]["gui_active_cycles"])[:12]:
    print(f"{name:40s} n={row['launches']:5d} mfma_util {row['mfma_util']:.3f}")
