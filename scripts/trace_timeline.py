"""Timeline of the LAST training iteration in a rocprofv3 kernel trace:
python scripts/trace_timeline.py <kernel_trace.csv> [launches per iteration marker kernel]
An iteration is delimited by the add-noise kernel (first launch of an iteration).  Prints, per queue, the busy time, and
every launch with its start offset, duration and queue, so that overlap between the chains can be read off."""
import csv, sys, collections

rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else "add_noise"
for r in rows:
    r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
marks = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
if len(marks) < 3:
    sys.exit(f"fewer than 3 launches of a kernel named *{marker}*")
lo, hi = marks[-3], marks[-2]
it = rows[lo:hi]
t0 = it[0]["s"]
print(f"iteration: {len(it)} launches, {(rows[hi]['s'] - t0) / 1e3:.1f} us between markers")
busy = collections.defaultdict(int)
for r in it:
    busy[r.get("Queue_Id", "?")] += r["e"] - r["s"]
for q, b in sorted(busy.items()):
    print(f"queue {q}: busy {b / 1e3:.1f} us")
short = lambda n: n.replace("void ", "").replace("(anonymous namespace)::", "")[:90]
for r in it:
    print(f"{(r['s'] - t0) / 1e3:9.1f} {(r['e'] - r['s']) / 1e3:8.1f} q{r.get('Queue_Id', '?'):>3} grid {r.get('Grid_Size_X', r.get('Grid_Size', '?')):>8} {short(r['Kernel_Name'])}")
