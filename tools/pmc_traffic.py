"""HBM traffic of the GEMM kernel from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, one pass each: they do
not fit one pass on gfx950) over `bench.py --no-graph --lanes 1 --steps 1 --warmup 0`.

bench.py ends with measure_gemm_roofline(): 1 + 3 repetitions of the GEMM launches of one forward at the full row
count (45 with the fused LayerNorm form, 33 without); the LAST 3 x launches gemm dispatches of each pass are those.  Per launch:
    bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024
(FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B, so it is doubled:
MI355X_MICROARCH.md, HBM section).  Writes profiles/gemm_traffic.json {"<dtype>": bytes per launch, ...}.

usage: python tools/pmc_traffic.py <fetch_dir> <write_dir> <dtype> <rows_per_launch> [out.json] [launches=45]"""
import csv
import glob
import json
import os
import sys


def last_gemm(dirname, counter, n=99):
    vals = []
    for f in sorted(glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True)):
        rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter and "gemm" in r["Kernel_Name"]]
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
        vals += [(r["Kernel_Name"], float(r["Counter_Value"])) for r in rows]
    assert len(vals) >= n, (dirname, counter, len(vals))
    return vals[-n:]


def main():
    fdir, wdir, dtype, rows = sys.argv[1:5]
    out = sys.argv[5] if len(sys.argv) > 5 else os.path.join(os.path.dirname(__file__), "..", "profiles", "gemm_traffic.json")
    n = 3 * (int(sys.argv[6]) if len(sys.argv) > 6 else 45)
    fetch, write = last_gemm(fdir, "FETCH_SIZE", n), last_gemm(wdir, "WRITE_SIZE", n)
    f_kib = sum(v for _, v in fetch) / len(fetch)
    w_kib = sum(v for _, v in write) / len(write)
    per_launch = (2.0 * f_kib + w_kib) * 1024.0
    rec = json.load(open(out)) if os.path.isfile(out) else {}
    if rec.get("rows_per_launch") != int(rows) or rec.get("launches_per_slab") != n // 3:
        rec = {"rows_per_launch": int(rows), "launches_per_slab": n // 3}     # figures of another slab size / launch sequence do not mix
    rec[dtype] = round(per_launch)
    rec[dtype + "_detail"] = {"fetch_kib_raw_avg": round(f_kib, 1), "write_kib_avg": round(w_kib, 1),
                              "launches_averaged": len(fetch), "formula": "(2*FETCH_SIZE + WRITE_SIZE)*1024"}
    json.dump(rec, open(out, "w"), indent=1, sort_keys=True)
    # bench.py reads this copy: profiles/ is not sent to the GPU box
    json.dump(rec, open(os.path.join(os.path.dirname(__file__), "..", "bench_traffic.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps(rec[dtype + "_detail"]), "->", rec[dtype], "bytes per launch")


if __name__ == "__main__":
    main()
