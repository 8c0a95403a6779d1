"""MFMA utilisation of the GEMM kernel from one rocprofv3 PMC pass (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE) over
`bench.py --no-graph --lanes 1 --steps 1 --warmup 0`: the LAST 3 x 45 gemm dispatches (3 x 33 without the fused LayerNorm form) are the roofline launches.
    busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)
usage: python tools/mfma_busy.py <mfma_dir> [n=135]"""
import csv
import glob
import os
import sys

d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 135
rows = []
for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
    rows += [r for r in csv.DictReader(open(f)) if "gemm" in r["Kernel_Name"]]
by = {}
for r in rows:
    by.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
    by[int(r["Dispatch_Id"])]["dur"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
ids = sorted(by)[-n:]
busy = sum(by[i]["SQ_VALU_MFMA_BUSY_CYCLES"] for i in ids)
act = sum(by[i]["GRBM_GUI_ACTIVE"] for i in ids)
dur = sum(by[i]["dur"] for i in ids)
print(f"{len(ids)} GEMM dispatches: MFMA busy {busy:.4g} cycles, GRBM_GUI_ACTIVE {act:.4g} (sum over 8 XCDs), "
      f"busy fraction {busy / (act / 8 * 1024):.4f}; mean clock over the dispatches {act / 8 / dur:.3f} GHz "
      f"(profiled pass; average duration {dur / len(ids) / 1e3:.1f} us)")
