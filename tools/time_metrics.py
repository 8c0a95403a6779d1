"""Wall time of the on-device per-gene metrics (Spearman by counting; moments + JS passes) at two sizes."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from stdiff_amd import synth as O
from stdiff_amd.metrics import gene_metrics, spearman
for N, G in ((10000, 500), (20000, 1000)):
    raw = torch.from_numpy(O.synth_expression(N, G, 4).astype(np.float32)).cuda()
    imp = (raw + 0.1 * torch.randn_like(raw)).clamp_min(0) + 1e-3
    spearman(imp, raw); torch.cuda.synchronize()
    t0 = time.perf_counter(); r = spearman(imp, raw); torch.cuda.synchronize(); t1 = time.perf_counter()
    m = gene_metrics(imp, raw, spcc=False); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"N={N} G={G}: spearman {1e3*(t1-t0):.1f} ms ({N*N*G*2/(t1-t0)/1e12:.2f} T pair-compares/s), other metrics {1e3*(t2-t1):.1f} ms; median rho {np.nanmedian(r):.4f} PCC {np.nanmedian(m['PCC']):.4f}", flush=True)
