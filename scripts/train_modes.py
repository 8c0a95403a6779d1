"""Milliseconds per training iteration (512 cells, bf16) by enqueue mode: Python launches one by one / one C-ABI enqueue
for forward and backward each, eager with and without the weight-gradient fork, and the whole iteration as one hipGraph.
python scripts/train_modes.py [genes] [hidden]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from stdiff_amd import synth as O
from stdiff_amd.train_engine import TrainEngine
g = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
h = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dev = torch.device("cuda:0")
st = torch.from_numpy(O.synth_expression(8 * 512, g, 1) * 2 - 1).to(dev)
for label, c_enq, side, graph in (("python enqueue, eager", False, "1", False), ("python enqueue, eager + fork", False, "2", False),
                                  ("C enqueue, eager", True, "1", False), ("C enqueue, eager + fork", True, "2", False),
                                  ("python enqueue, hipGraph + fork", False, "1", True), ("C enqueue, hipGraph + fork", True, "1", True)):
    os.environ["STDIFF_TRAIN_SIDE_STREAM"] = side
    eng = TrainEngine(B.build_model("bf16", dev, g, h), lr=1e-4, diffusion_step=1000, mask=O.kfold_mask(g, 0))
    eng.c_enqueue = c_enq
    for i in range(8):
        eng.step(st[:512], st[:512], use_graph=graph)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 40
    for i in range(n):
        lo = (i % 8) * 512
        eng.step(st[lo:lo + 512], st[lo:lo + 512], use_graph=graph)
    t_host = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{label:36s} {dt * 1e3:7.3f} ms/step  ({512 / dt / 1e3:6.1f} k cells/s; host enqueue {t_host * 1e3:6.3f} ms)", flush=True)
    del eng
    torch.cuda.empty_cache()
