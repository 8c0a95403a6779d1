"""One training configuration, hipGraph mode, for kernel traces: python scripts/train_only.py [genes] [hidden] [steps] [graph 0|1]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from stdiff_amd import synth as O
from stdiff_amd.train_engine import TrainEngine
g = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
h = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
n = int(sys.argv[3]) if len(sys.argv) > 3 else 30
graph = (sys.argv[4] if len(sys.argv) > 4 else "1") == "1"
dev = torch.device("cuda:0")
st = torch.from_numpy(O.synth_expression(8 * 512, g, 1) * 2 - 1).to(dev)
lazy = os.environ.get("LAZY", "1") == "1"
join = os.environ.get("JOIN", "0") == "1"          # 0: step(join=False), what normal_train_stDiff and bench.py use
eng = TrainEngine(B.build_model("bf16", dev, g, h), lr=1e-4, diffusion_step=1000, mask=O.kfold_mask(g, 0), lazy_update=lazy)
for i in range(8):
    eng.step(st[:512], st[:512], use_graph=graph, join=join)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(n):
    lo = (i % 8) * 512
    eng.step(st[lo:lo + 512], st[lo:lo + 512], use_graph=graph, join=join)
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) / n * 1e3:.3f} ms/step, graph={graph}, lazy={lazy}, join={join}")
eng.finish()
