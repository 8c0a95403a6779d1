"""GPU: BASELINE.json configs[4] at its full size (10 000 spots x 500 genes x 5 folds x the whole 1000-step loop).

Collected BEFORE the tests that start child GPU processes (tests/test_gpu_host_c.py, the two-rank data-parallel test of
tests/test_gpu_train.py) on purpose: until round 3 this test's fifteen driver calls left the process in a state in which
no new GPU process could start (every sampler drew fresh HIP streams; DESIGN.md 6a'), and the file was named to sort
last.  With the per-device stream cache (engine.lane_streams) the children start normally after it --
test_child_gpu_process_starts_after_the_folds checks exactly that."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import cpu_ref as O  # noqa: E402

from test_gpu_baseline_sizes import DEV, FP32_TOL, relerr  # noqa: E402


def test_config4_five_folds_full_size_tolerance_sweep(progress):
    """configs[4] at its full size: 10 000 spots x 500 genes, hidden 1024, chunks of 512 (the last one 272 rows), all
    FIVE gene folds through the driver flow (per-fold weights loaded as checkpoints, then the whole 1000-step reverse loop) in fp32,
    bf16 and fp16 with the same weights and the same Philox noise.  Reported and asserted per fold on the imputed
    (held-out) genes: per-gene Pearson / cosine of each 16-bit result against the fp32 one.  The oracle pins the fp32
    run itself: one chunk per fold, teacher-forced on the run's own final state, within 1e-5 (max-norm)."""
    from stdiff_amd import engine as E
    from stdiff_amd.driver import diffusion_impute, fold_mask, kfold_gene_splits
    from stdiff_amd.metrics import gene_metrics
    from stdiff_amd.model.stDiff_model import DiT_stDiff
    N, G, H, depth, heads, B, T = 10000, 500, 1024, 6, 16, 512, 1000
    seq = O.synth_expression(4096, G, 21)
    st = O.synth_expression(N, G, 22)
    import tempfile
    with tempfile.TemporaryDirectory() as ck:
        kw = dict(lr=0.00016046744893538737, depth=depth, num_epoch=2, diffusion_step=T, batch_size=B, hidden_size=H,
                  head=heads, device=torch.device(DEV), save_dir=ck, sample_seed=4242)
        # a model two epochs from the reference initialisation still outputs ~0 (zero-initialised adaLN / output
        # layers), which would make every precision agree trivially: the folds load re-randomised weights through the
        # driver's checkpoint path instead (training at this shape: test_config2_* and tests/test_gpu_train.py)
        for k in range(5):
            torch.save(O.make_params(G, H, depth, 60 + k), f"{ck}/stDiff{k + 1}.pt")
        res = {}
        splits = kfold_gene_splits(G, 5, 0)
        for dt in ("bf16", "fp16", "fp32"):
            res[dt] = np.zeros((N, G), dtype=np.float32)
            for k, (_, te) in enumerate(splits):     # fold by fold, with a progress line each (an fp32 fold takes 20 s)
                res[dt][:, te] = diffusion_impute(seq, st, compute_dtype=dt, folds=[k], **kw)[:, te]
                progress(f"configs[4] {dt} fold {k + 1}/5 done")
        for k, (_, te) in enumerate(splits):
            ref = torch.from_numpy(res["fp32"][:, te]).to(DEV)
            line = [f"configs[4] fold {k + 1} ({len(te)} held-out genes):"]
            for dt, floor in (("bf16", 0.999), ("fp16", 0.9999)):
                gm = gene_metrics(torch.from_numpy(res[dt][:, te]).to(DEV), ref, js=False, spcc=False)
                a, b = res[dt][:, te].astype(np.float64), res["fp32"][:, te].astype(np.float64)
                line.append(f"{dt} Pearson min {gm['PCC'].min():.6f} median {np.median(gm['PCC']):.7f} cosine min "
                            f"{gm['cosine'].min():.6f} rel-l2 {np.linalg.norm(a - b) / np.linalg.norm(b):.2e};")
                assert gm["PCC"].min() > floor and gm["cosine"].min() > floor, (k, dt)
            print(" ".join(line))
            assert np.isfinite(res["fp32"]).all()
            # the oracle on one chunk of this fold's model (the ragged last chunk for fold 5), teacher-forced
            sd = torch.load(f"{ck}/stDiff{k + 1}.pt", map_location="cpu", weights_only=True)
            m = DiT_stDiff(G, H, depth, 'dit', heads, 6, compute_dtype="fp32")
            m.load_state_dict(sd)
            m = m.to(DEV).eval()
            lo, hi = (19 * B, N) if k == 4 else (k * B, (k + 1) * B)
            mask = fold_mask(G, te)
            x_c = torch.from_numpy((res["fp32"][lo:hi] * 2 - 1).astype(np.float32))     # a state of the right scale
            y_c = torch.from_numpy(((st[lo:hi] * mask) * 2 - 1).astype(np.float32))
            tt = torch.full((hi - lo,), 37 * (k + 1), dtype=torch.int64)
            with torch.no_grad():
                got = m(x_c, tt, y_c).cpu()
                want = O.dit_forward({n: v.float() for n, v in sd.items()}, x_c, tt, y_c, heads, depth)
            e = relerr(got, want)
            print(f"   fp32 denoiser call on chunk rows {lo}:{hi} vs oracle: {e:.2e}")
            assert e < FP32_TOL, k


def test_child_gpu_process_starts_after_the_folds():
    """DESIGN.md 6a': after the fifteen sampling runs above a NEW GPU process must still start (hsa_init in the child
    used to die with SIGSEGV).  Runs in the same pytest process, right after the folds."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, "-c", "import torch; print(float(torch.ones(4, device='cuda:0').sum()))"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("4.0"), (r.returncode, r.stdout[-200:], r.stderr[-500:])
