"""Training iteration through TrainEngine.step() against the replay of its hipGraph alone (no input staging, no noise /
timestep draw between replays): python scripts/graph_only.py <genes> <hidden> <steps>"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from stdiff_amd import synth as O
from stdiff_amd.train_engine import TrainEngine
g = int(sys.argv[1]); h = int(sys.argv[2]); n = int(sys.argv[3])
dev = torch.device("cuda:0")
st = torch.from_numpy(O.synth_expression(8 * 512, g, 1) * 2 - 1).to(dev)
eng = TrainEngine(B.build_model("bf16", dev, g, h), lr=1e-4, diffusion_step=1000, mask=O.kfold_mask(g, 0), lazy_update=True)
for i in range(8):
    eng.step(st[:512], st[:512], use_graph=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(n):
    lo = (i % 8) * 512
    eng.step(st[lo:lo + 512], st[lo:lo + 512], use_graph=True)
torch.cuda.synchronize()
full = (time.perf_counter() - t0) / n * 1e3
gr = eng._graphs[512][0]["graph"]
with torch.cuda.stream(eng._stream):
    for i in range(3): gr.launch()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n): gr.launch()
    torch.cuda.synchronize()
only = (time.perf_counter() - t0) / n * 1e3
print(f"step() {full:.3f} ms   graph replay alone {only:.3f} ms")
eng.finish()
