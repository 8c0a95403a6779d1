"""normal_train_stDiff end to end (the public training call) with the host DataLoader of the reference and with the
device-resident loader: seconds per epoch and per iteration.  python scripts/train_loop_time.py [cells] [genes] [hidden] [epochs]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from stdiff_amd import synth as O
from stdiff_amd.process.data import get_data_loader
from stdiff_amd.model.stDiff_train import normal_train_stDiff
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20480
g = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
h = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
ep = int(sys.argv[4]) if len(sys.argv) > 4 else 3
dev = torch.device("cuda:0")
x = (O.synth_expression(n, g, 1) * 2 - 1).astype(np.float32)
mask = O.kfold_mask(g, 0)
xc = x * mask
for name, kw in (("host DataLoader", {}), ("device loader", {"device": dev})):
    model = B.build_model("bf16", dev, g, h)
    dl = get_data_loader(x, xc, batch_size=512, is_shuffle=True, **kw)
    normal_train_stDiff(model, dl, lr=1e-4, num_epoch=1, device=dev, is_tqdm=False, mask=mask)      # warm-up (graph capture)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    normal_train_stDiff(model, dl, lr=1e-4, num_epoch=ep, device=dev, is_tqdm=False, mask=mask)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    its = ep * ((n + 511) // 512)
    print(f"{name}: {dt / ep:.3f} s per epoch of {n} cells, {dt / its * 1e3:.3f} ms per iteration", flush=True)
