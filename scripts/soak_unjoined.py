"""Soak of TrainEngine.step(join=False): N iterations at the bench's training shape with fresh input tensors per step
(dropped right after the call), against the same N iterations with join=True -- parameters must be equal bit for bit
(the iteration is deterministic): python scripts/soak_unjoined.py [steps] [genes] [hidden]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from stdiff_amd import synth as O
from stdiff_amd.train_engine import TrainEngine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
g = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
h = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
dev = torch.device("cuda:0")
host = torch.from_numpy(O.synth_expression(16 * 512, g, 1) * 2 - 1)
out = {}
for join in (True, False):
    torch.manual_seed(7)
    eng = TrainEngine(B.build_model("bf16", dev, g, h), lr=1e-4, diffusion_step=1000, mask=O.kfold_mask(g, 0), lazy_update=True)
    total = torch.zeros(1, device=dev)
    t0 = time.perf_counter()
    for i in range(n):
        lo = (i % 16) * 512
        x = host[lo:lo + 512].to(dev, non_blocking=True)          # a fresh device tensor per step
        xc = x * 0.5
        loss, _ = eng.step(x, xc, use_graph=True, join=join)
        eng.accumulate(total, loss)
        del x, xc
    eng.finish()
    torch.cuda.synchronize()
    out[join] = (eng.flat_p.clone(), eng.m.clone(), float(total), (time.perf_counter() - t0) / n * 1e3)
    print(f"join={join}: {out[join][3]:.3f} ms per step (upload included), loss sum {out[join][2]:.6f}", flush=True)
same = torch.equal(out[True][0], out[False][0]) and torch.equal(out[True][1], out[False][1])
print("parameters and moments bit-equal:", same, " loss sums:", out[True][2], out[False][2])
sys.exit(0 if same else 1)
