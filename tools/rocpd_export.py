"""rocprofv3 (ROCm 7.2) writes a rocpd SQLite database unless --output-format csv is given.  This turns one into the
two CSV summaries kept under profiles/:
    python tools/rocpd_export.py stats    <results.db> <out.csv>    # what --kernel-trace --stats prints per kernel
    python tools/rocpd_export.py counters <results.db> <out_counter_collection.csv>   # one row per dispatch x counter
The counter CSV carries the columns tools/pmc_traffic.py reads."""
import csv
import sqlite3
import statistics
import sys


def stats(db, out):
    c = sqlite3.connect(db)
    per = {}
    for name, dur in c.execute("select name, duration from kernels"):
        per.setdefault(name, []).append(dur)
    total = sum(sum(v) for v in per.values())
    with open(out, "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for name, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([name, len(v), sum(v), round(sum(v) / len(v), 6), round(100.0 * sum(v) / total, 2), min(v), max(v),
                        round(statistics.pstdev(v), 6)])


def counters(db, out):
    c = sqlite3.connect(db)
    q = ("select dispatch_id, kernel_name, grid_size, workgroup_size, lds_block_size, scratch_size, vgpr_count, "
         "accum_vgpr_count, sgpr_count, counter_name, value, start, end from counters_collection order by dispatch_id")
    with open(out, "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size",
                    "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Counter_Name", "Counter_Value", "Start_Timestamp",
                    "End_Timestamp"])
        for r in c.execute(q):
            w.writerow(r)


if __name__ == "__main__":
    {"stats": stats, "counters": counters}[sys.argv[1]](sys.argv[2], sys.argv[3])
