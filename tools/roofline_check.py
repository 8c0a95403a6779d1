"""Cross-check of bench.py's event-timed GEMM launch average against the rocprofv3 kernel trace of the same command.
usage: python tools/roofline_check.py <kt_results.db> [launches=33] [reps=3]
The roofline launches are the last reps*launches GEMM dispatches before the training leg starts (its first kernel is
add_noise_splice); split-K reduces are not GEMM launches."""
import sqlite3
import sys

db = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 33
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
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
