"""HIP-event time and TFLOP/s of each of the 8 distinct GEMM shapes of one denoiser forward at R rows (default 20000),
single stream: python tools/gemm_layers.py [R]"""
import sys, torch
sys.path.insert(0, '.')
import bench as B
from stdiff_amd import engine as E
dev = torch.device('cuda:0')
model = B.build_model("bf16", dev)
eng = model.engine()
R = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
lens = [min(512, R - i) for i in range(0, R, 512)]
ws = eng.workspace(lens)
L = B.gemm_launches(eng, ws)
for _, _, _, f in L: f()
torch.cuda.synchronize()
names = ["in"] + ["ada", "qkv", "proj", "fc1", "fc2"] * 6 + ["fada", "out"]
tot = {}
for rep in range(5):
    evs = []
    for m, n, k, f in L:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); evs.append((a, b))
    torch.cuda.synchronize()
    for (m, n, k, f), (a, b), nm in zip(L, evs, names):
        t = tot.setdefault(nm, [0.0, 0, 0.0]); t[0] += a.elapsed_time(b); t[1] += 1; t[2] = 2.0 * m * n * k
s = 0
for nm, (ms, cnt, fl) in tot.items():
    us = ms / cnt * 1e3
    per_fwd = us * (6 if nm in ("ada", "qkv", "proj", "fc1", "fc2") else 1)
    s += per_fwd
    print(f"{nm:5s} {us:8.1f} us  {fl/us/1e6:7.0f} TF   per forward {per_fwd:8.1f} us")
print("sum per forward", s)
