"""HIP-event time and TFLOP/s of each of the 8 distinct GEMM shapes of one denoiser forward at R rows (default 65536 =
one row slab of the sampler), single stream: python scripts/gemm_layers.py [R] [dtype]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
dev = torch.device('cuda:0')
R = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dtype = sys.argv[2] if len(sys.argv) > 2 else "bf16"
model = B.build_model(dtype, dev)
eng = model.engine()
lens = [min(512, R - i) for i in range(0, R, 512)]
ws = eng.workspace(lens)
L = B.gemm_launches(eng, ws)
for buf in (ws.xp, ws.s, ws.xn, ws.o, ws.u, ws.h, ws.gates if ws.fused else ws.mod):
    buf.normal_(0, 0.5)
import time
t0 = time.time()
while time.time() - t0 < 2.0:          # a fresh box ramps its clocks for a second or two
    for _, _, _, f, _ in L: f()
    torch.cuda.synchronize()
names = ["in"] + (["gate", "ln1", "qkv", "proj", "ln2", "fc1", "fc2"] if ws.fused else ["ada", "qkv", "proj", "fc1", "fc2"]) * 6 + ["fada", "out"]
tot = {}
for rep in range(5):
    evs = []
    for m, n, k, f, _ in L:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); evs.append((a, b))
    torch.cuda.synchronize()
    for (m, n, k, f, _), (a, b), nm in zip(L, evs, names):
        t = tot.setdefault(nm, [0.0, 0, 0.0]); t[0] += a.elapsed_time(b); t[1] += 1; t[2] = 2.0 * m * n * k
s = fl_all = 0
for nm, (ms, cnt, fl) in tot.items():
    us = ms / cnt * 1e3
    k = 6 if nm in ("ada", "qkv", "proj", "fc1", "fc2", "gate", "ln1", "ln2") else 1
    s += us * k; fl_all += fl * k
    print(f"{nm:5s} {us:8.1f} us  {fl/us/1e6:7.0f} TF   per forward {us * k:8.1f} us")
print(f"sum per forward {s:.1f} us = {fl_all / s / 1e6:.0f} TF  (R={R}, {dtype}, env TILE={os.environ.get('STDIFF_GEMM_TILE')})")
