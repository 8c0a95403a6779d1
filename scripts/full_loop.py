"""SURVEY 8(d)(i) as written: imputed spots/s = N / wall time of ONE COMPLETE T = 1000 reverse loop through the public
call (sample_stDiff: H2D of the condition / gt matrices, x_T init, 1000 graph replays, D2H of the [N, G] result), at the
BASELINE configs[3] size on one GPU.  python scripts/full_loop.py [spots] [dtype]"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from stdiff_amd import synth as O
from stdiff_amd.model.sample import sample_stDiff
from stdiff_amd.model.stDiff_scheduler import NoiseScheduler
from stdiff_amd.process.data import get_data_loader
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dtype = sys.argv[2] if len(sys.argv) > 2 else "fp16"
dev = torch.device("cuda:0")
G = B.G
model = B.build_model(dtype, dev)
mask = O.kfold_mask(G, 0)
gt = O.synth_expression_device(N, G, 1, dev)
gt = (gt * torch.from_numpy(mask).to(dev) * 2 - 1).cpu().numpy()           # host matrix, as a driver would hold it
print(f"inputs ready: gt {gt.shape} {gt.nbytes / 1e9:.1f} GB on the host", flush=True)
for rep in range(1):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dl = get_data_loader(gt, gt, batch_size=512, is_shuffle=False, device=dev)       # H2D, device-resident
    t1 = time.perf_counter()
    out = sample_stDiff(model, dl, NoiseScheduler(1000, beta_schedule="cosine"), mask=mask, gt=gt, device=dev, num_step=1000,
                        sample_shape=(N, G), is_condi=True, sample_intermediate=1000, model_pred_type='noise', seed=7,
                        is_tqdm=True)
    t2 = time.perf_counter()
    known = mask.astype(bool)
    ok = bool(np.isfinite(out[::997]).all()) and np.array_equal(out[::997][:, known], gt[::997][:, known])
    print(f"\nfull loop {dtype}: loader {t1 - t0:.2f} s + sample_stDiff (H2D, init, 1000 steps, D2H) {t2 - t1:.2f} s = "
          f"{t2 - t0:.2f} s -> {N / (t2 - t0):.1f} spots/s end to end ({N / (t2 - t1):.1f} without the loader); "
          f"finite and known genes bit-equal on a row sample: {ok}", flush=True)
