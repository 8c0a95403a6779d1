"""Cross-check of bench.py's event-timed GEMM launch average against the rocprofv3 kernel trace of the same command.
usage: python tools/roofline_check.py <kt_results.db | kt_kernel_trace.csv> [launches=45] [reps=3]
The roofline launches are the last reps*launches GEMM dispatches before the training leg starts (its first kernel is
add_noise_splice); split-K reduces are not GEMM launches."""
import sqlite3
import sys

db = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 45
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
if db.endswith(".csv"):                      # rocprofv3 --output-format csv: *_kernel_trace.csv
    import csv
    tr = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(db))]
    t_train = min([s for k, s, e in tr if "add_noise" in k] or [1 << 62])
    rows = sorted((s, k, e - s) for k, s, e in tr if "gemm" in k and "splitk" not in k and s < t_train)
    rows = [(k, d) for _, k, d in rows][-n * reps:]
else:
    c = sqlite3.connect(db)
    t_train = c.execute("select min(start) from kernels where name like '%add_noise%'").fetchone()[0] or 1 << 62
    rows = list(c.execute("select name, duration from kernels where name like '%gemm%' and name not like '%splitk%' "
                          "and start < ? order by start", (t_train,)))[-n * reps:]
by = {}
for name, d in rows:
    by.setdefault(name.split("(anonymous namespace)::")[1].split("(")[0], []).append(d)
print(f"{len(rows)} roofline GEMM dispatches, average {sum(d for _, d in rows) / len(rows) / 1e3:.1f} us")
for k, v in by.items():
    print(f"  {k}: {len(v)} launches, average {sum(v) / len(v) / 1e3:.1f} us")
