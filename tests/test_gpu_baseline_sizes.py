"""GPU: the hot path at BASELINE.json's own shapes.

configs[0] (4096 cells x 256 genes, hidden 512, 16 heads -> d = 32, chunks of 2048) is small enough for the oracle:
one teacher-forced reverse step and one training iteration are compared element by element.
configs[1] (20 000 spots x 1000 genes, hidden 1024, chunks of 512, last chunk 32 rows) is checked through
size-independent properties of the reverse loop plus the oracle on single chunks picked from the full run:
  * known genes stay bit-equal to gt after every step (mask splice, sample.py:93-94),
  * a chunk's result does not depend on the other chunks or on the lane count (attention is per chunk; noise is
    keyed by the global row): chunk k run alone == chunk k inside the full run, bit for bit,
  * the first chunk, the last full chunk and the ragged last chunk of the full fp32 run match the oracle's denoiser
    + scheduler step to the fp32 tolerance."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import cpu_ref as O  # noqa: E402

DEV = "cuda:0"
FP32_TOL = 1e-5
BF16_TOL = 3e-2


def relerr(a, b):
    a, b = torch.as_tensor(a).float().cpu(), torch.as_tensor(b).float().cpu()
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30)


def build(G, H, depth, heads, seed, dtype):
    from stdiff_amd.model.stDiff_model import DiT_stDiff
    m = DiT_stDiff(G, H, depth, 'dit', heads, 6, compute_dtype=dtype)
    p = O.make_params(G, H, depth, seed)
    m.load_state_dict(p, strict=True)
    return m.to(DEV), p


def problem(N, G, fold=0, seed=1):
    mask = O.kfold_mask(G, fold)
    gt = ((O.synth_expression(N, G, seed) * mask) * 2 - 1).astype(np.float32)
    return mask, gt


# ---------------------------------------------------------------- configs[0]
@pytest.mark.parametrize("dtype", ["fp32", "bf16", "fp16"])
def test_config0_reverse_step_against_oracle(dtype):
    from stdiff_amd import engine as E
    from stdiff_amd.model.stDiff_scheduler import NoiseScheduler
    N, G, H, depth, heads, B, T = 4096, 256, 512, 6, 16, 2048, 1000
    m, p = build(G, H, depth, heads, 11, dtype)
    m.eval()
    mask, gt = problem(N, G)
    t = 700
    gen = torch.Generator().manual_seed(5)
    x_t = torch.randn(N, G, generator=gen)
    x_t = x_t * (1 - torch.from_numpy(mask)) + torch.from_numpy(gt) * torch.from_numpy(mask)
    z = torch.randn(N, G, generator=gen)
    tb = O.make_tables(T, schedule="cosine")
    with torch.no_grad():
        ref_eps = O.denoise_chunked(p, x_t, t, torch.from_numpy(gt), B, heads, depth)
        ref_x, _ = O.step(tb, ref_eps, t, x_t, z)
        ref_x = ref_x * (1 - torch.from_numpy(mask)) + torch.from_numpy(gt) * torch.from_numpy(mask)
        gtd, md = torch.from_numpy(gt).to(DEV), torch.from_numpy(mask).to(DEV)
        smp = E.ReverseSampler(m.engine(), NoiseScheduler(T, beta_schedule="cosine"), [B, B], gtd, gtd, md,
                               use_graph=False)
        smp.set_x(x_t.to(DEV))
        eps = smp.denoise_only(t).clone()
        smp.set_x(x_t.to(DEV))
        zz = torch.zeros(1, N, G, device=DEV)
        zz[0] = z.to(DEV)
        out = smp.run([t], z=zz).clone()
    e_eps, e_x = relerr(eps, ref_eps), relerr(out, ref_x)
    print(f"configs[0] {dtype}: eps rel err {e_eps:.3e}, x_(t-1) rel err {e_x:.3e}")
    tol = {"fp32": FP32_TOL, "bf16": BF16_TOL, "fp16": 1e-3}[dtype]
    assert e_eps < tol and e_x < tol
    known = mask.astype(bool)
    assert np.array_equal(out.cpu().numpy()[:, known], gt[:, known])


def test_config0_training_iteration_against_oracle():
    """One reference iteration on a batch of 2048 cells at configs[0]'s model (fp32): loss, pre-clip gradient norm,
    every gradient and the parameters after clip + AdamW."""
    from stdiff_amd.train_engine import TrainEngine
    G, H, depth, heads, C, T = 256, 512, 6, 16, 2048, 1000
    m, p = build(G, H, depth, heads, 12, "fp32")
    mask, _ = problem(8, G)
    st = (O.synth_expression(C, G, 2) * 2 - 1).astype(np.float32)
    gen = torch.Generator().manual_seed(9)
    noise = torch.randn(C, G, generator=gen)
    tt = torch.randint(1, T, (C,), generator=gen)
    x = torch.from_numpy(st)
    xc = torch.from_numpy(st * mask * 1.0).float()
    tb = O.make_tables(T, schedule="cosine")
    pr = {k: v.clone() for k, v in p.items()}
    loss_r, gn_r, raw = O.train_step(pr, tb, {}, x, xc, noise, tt, mask, 1e-4, heads, depth)
    te = TrainEngine(m, lr=1e-4, diffusion_step=T, mask=mask)
    loss, gn = te.step(x.to(DEV), xc.to(DEV), noise=noise.to(DEV), timesteps=tt.to(DEV))
    assert abs(float(loss) - float(loss_r)) < 1e-5 * max(1.0, float(loss_r))
    assert abs(float(gn) - float(gn_r)) < 1e-4 * float(gn_r)
    gscale = max(float(g.abs().max()) for g in raw.values())
    for n, g in raw.items():
        assert float((te.g[n].cpu() - g).abs().max()) < 1e-4 * gscale, n
    sd = m.state_dict()
    for n in ("blks.0.attn.proj.weight", "blks.5.mlp.fc2.weight", "in_layer.0.weight", "out_layer.linear.weight"):
        # the first Adam step moves every weight by ~lr * sign(g): compare the UPDATE to 2 % of lr; where |g| is at the
        # level of fp32 summation noise Adam's normalisation amplifies it, so a small share of elements may differ
        diff = (sd[n].cpu() - pr[n]).abs()
        assert float(diff.max()) <= 2.1e-4, n
        assert float((diff > 2e-6).float().mean()) < 1e-3, (n, float((diff > 2e-6).float().mean()))


# ---------------------------------------------------------------- configs[1]
def _config1(dtype):
    N, G, H, depth, heads, B, T = 20000, 1000, 1024, 6, 16, 512, 1000
    m, p = build(G, H, depth, heads, 1, dtype)
    m.eval()
    mask, gt = problem(N, G)
    lens = [b - a for a, b in O.chunk_bounds(N, B)]
    return (N, G, H, depth, heads, B, T), m, p, mask, gt, lens


def test_config1_full_size_properties_bf16():
    from stdiff_amd import engine as E
    from stdiff_amd.model.stDiff_scheduler import NoiseScheduler
    (N, G, H, depth, heads, B, T), m, p, mask, gt, lens = _config1("bf16")
    assert len(lens) == 40 and lens[-1] == 32
    gtd, md = torch.from_numpy(gt).to(DEV), torch.from_numpy(mask).to(DEV)
    ns = NoiseScheduler(T, beta_schedule="cosine")
    ts = list(range(T))[::-1][:4]
    known = torch.from_numpy(mask.astype(bool)).to(DEV)
    with torch.no_grad():
        full = E.LanedSampler(m.engine(), ns, lens, gtd, gtd, md, seed=1234, lanes=2)
        full.init()
        out = full.run(ts[:1])
        assert torch.equal(out[:, known], gtd[:, known])
        out = full.run_more(len(ts) - 1).clone()
        torch.cuda.synchronize()
        assert bool(torch.isfinite(out).all())
        assert torch.equal(out[:, known], gtd[:, known])                 # splice: known genes bit-equal to gt
        assert float((out[:, ~known] - gtd[:, ~known]).abs().max()) > 0  # unknown genes are being sampled
        one = E.LanedSampler(m.engine(), ns, lens, gtd, gtd, md, seed=1234, lanes=1)
        one.init()
        assert torch.equal(one.run(ts), out)                             # lane count does not change a bit
        for k in (0, 17, 39):                                            # chunk alone == chunk in the full run
            lo = k * B
            hi = lo + lens[k]
            s = E.ReverseSampler(m.engine(), ns, [lens[k]], gtd[lo:hi], gtd[lo:hi], md, seed=1234, row_offset=lo,
                                 lane=7)
            s.init()
            assert torch.equal(s.run(ts), out[lo:hi]), k


def test_config1_chunks_against_oracle_fp32():
    """Teacher-forced reverse step of the full 20 000-row fp32 run; the oracle evaluates three of its 40 chunks."""
    from stdiff_amd import engine as E
    from stdiff_amd.model.stDiff_scheduler import NoiseScheduler
    (N, G, H, depth, heads, B, T), m, p, mask, gt, lens = _config1("fp32")
    gtd, md = torch.from_numpy(gt).to(DEV), torch.from_numpy(mask).to(DEV)
    t = 431
    gen = torch.Generator().manual_seed(6)
    mk = torch.from_numpy(mask)
    x_t = torch.randn(N, G, generator=gen) * (1 - mk) + torch.from_numpy(gt) * mk
    tb = O.make_tables(T, schedule="cosine")
    with torch.no_grad():
        smp = E.ReverseSampler(m.engine(), NoiseScheduler(T, beta_schedule="cosine"), lens, gtd, gtd, md,
                               use_graph=False)
        smp.set_x(x_t.to(DEV))
        eps = smp.denoise_only(t).cpu()
        for k in (0, 38, 39):
            lo = k * B
            hi = lo + lens[k]
            ref = O.dit_forward(p, x_t[lo:hi], torch.full((hi - lo,), t, dtype=torch.int64), torch.from_numpy(gt[lo:hi]),
                                heads, depth)
            e = relerr(eps[lo:hi], ref)
            print(f"configs[1] fp32 chunk {k} ({hi - lo} rows): eps rel err {e:.3e}")
            assert e < FP32_TOL, k
            z = torch.randn(hi - lo, G, generator=gen)
            ref_x, _ = O.step(tb, ref, t, x_t[lo:hi], z)
            s = E.ReverseSampler(m.engine(), NoiseScheduler(T, beta_schedule="cosine"), [hi - lo], gtd[lo:hi], gtd[lo:hi],
                                 md, use_graph=False, lane=7)
            s.set_x(x_t[lo:hi].to(DEV))
            got = s.run([t], z=z.to(DEV)[None]).cpu()
            ref_x = ref_x * (1 - mk) + torch.from_numpy(gt[lo:hi]) * mk
            assert relerr(got, ref_x) < FP32_TOL, k


# ---------------------------------------------------------------- configs[2] (per-rank shape of the DDP run)
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_config2_shape_training_iteration_against_oracle(dtype):
    """configs[2]'s per-rank step: 512 cells x 2000 genes, hidden 2048 (16 heads -> d = 128), depth cut to 2 so that
    the oracle's autograd finishes in seconds.  fp32: loss, gradient norm and every gradient against the oracle;
    bf16: loss and gradient norm track the oracle (bf16 operands, fp32 accumulation)."""
    from stdiff_amd.train_engine import TrainEngine
    G, H, depth, heads, C, T = 2000, 2048, 2, 16, 512, 1000
    m, p = build(G, H, depth, heads, 15, dtype)
    mask, _ = problem(8, G)
    st = (O.synth_expression(C, G, 3) * 2 - 1).astype(np.float32)
    gen = torch.Generator().manual_seed(10)
    noise = torch.randn(C, G, generator=gen)
    tt = torch.randint(1, T, (C,), generator=gen)
    x = torch.from_numpy(st)
    xc = torch.from_numpy(st * mask * 1.0).float()
    tb = O.make_tables(T, schedule="cosine")
    loss_r, grads, _ = O.loss_and_grads(p, tb, x, xc, noise, tt, mask, heads, depth)
    # float64: torch's CPU fp32 norm over tensors of 10^7..10^8 elements is itself only good to ~1e-3 here
    gn_r = float(sum(g.double().pow(2).sum() for g in grads.values()) ** 0.5)
    te = TrainEngine(m, lr=1e-4, diffusion_step=T, mask=mask)
    loss, gn = te.step(x.to(DEV), xc.to(DEV), noise=noise.to(DEV), timesteps=tt.to(DEV))
    print(f"configs[2] shape {dtype}: loss {float(loss):.6f} (oracle {float(loss_r):.6f}), |g| {float(gn):.5f} (oracle {gn_r:.5f})")
    if dtype == "fp32":
        assert abs(float(loss) - float(loss_r)) < 1e-5 * max(1.0, float(loss_r))
        assert abs(float(gn) - gn_r) < 1e-4 * gn_r
        gscale = max(float(g.abs().max()) for g in grads.values())
        for n, g in grads.items():
            assert float((te.g[n].cpu() - g).abs().max()) < 1e-4 * gscale, n
    else:
        assert abs(float(loss) - float(loss_r)) < 2e-2 * float(loss_r)
        assert abs(float(gn) - gn_r) < 5e-2 * gn_r


# ---------------------------------------------------------------- configs[4]: fp32 vs bf16 / fp16 over a whole 1000-step loop
def test_config4_fp32_vs_bf16_full_loop():
    """configs[4]'s model (500 genes, hidden 1024, chunks of 512) on 2048 spots through the WHOLE 1000-step reverse loop
    in all three precisions with the same Philox noise: the bf16 and fp16 results are compared with the fp32 one per
    gene (Pearson and cosine over the spots, unknown genes only: the numbers the north star asks to report for the
    16-bit path), known genes are bit-equal in all."""
    from stdiff_amd.metrics import gene_metrics
    from stdiff_amd.model.sample import sample_stDiff
    from stdiff_amd.model.stDiff_scheduler import NoiseScheduler
    from stdiff_amd.process.data import get_data_loader
    N, G, H, depth, heads, B, T = 2048, 500, 1024, 6, 16, 512, 1000
    mask, gt = problem(N, G, fold=2, seed=7)
    res = {}
    for dtype in ("fp32", "bf16", "fp16"):
        m, _ = build(G, H, depth, heads, 4, dtype)
        m.eval()
        dl = get_data_loader(gt, gt, batch_size=B, is_shuffle=False, device=DEV)
        res[dtype] = sample_stDiff(m, dl, NoiseScheduler(T, beta_schedule="cosine"), mask=mask, gt=gt,
                                   device=torch.device(DEV), num_step=T, sample_shape=(N, G), is_condi=True,
                                   sample_intermediate=T, model_pred_type='noise', seed=99, is_tqdm=False)
        assert np.isfinite(res[dtype]).all()
    known = mask.astype(bool)
    assert np.array_equal(res["fp32"][:, known], gt[:, known]) and np.array_equal(res["bf16"][:, known], gt[:, known])
    unk = ~known
    gm = gene_metrics(torch.from_numpy(res["bf16"][:, unk]).to(DEV), torch.from_numpy(res["fp32"][:, unk]).to(DEV),
                      js=False, spcc=False)
    rel = np.abs(res["bf16"] - res["fp32"]).max() / np.abs(res["fp32"]).max()
    print(f"configs[4] full loop bf16 vs fp32: max rel diff {rel:.3e}; per-gene Pearson min {gm['PCC'].min():.5f} "
          f"median {np.median(gm['PCC']):.5f}; cosine min {gm['cosine'].min():.5f}")
    assert gm["PCC"].min() > 0.999 and gm["cosine"].min() > 0.999
    assert np.array_equal(res["fp16"][:, known], gt[:, known])
    g16 = gene_metrics(torch.from_numpy(res["fp16"][:, unk]).to(DEV), torch.from_numpy(res["fp32"][:, unk]).to(DEV),
                       js=False, spcc=False)
    a, b = res["fp16"][:, unk].astype(np.float64), res["fp32"][:, unk].astype(np.float64)
    rel16, l2 = np.abs(a - b).max() / np.abs(b).max(), np.linalg.norm(a - b) / np.linalg.norm(b)
    print(f"configs[4] full loop fp16 vs fp32: max rel diff {rel16:.3e}, rel-l2 {l2:.3e}; per-gene Pearson min "
          f"{g16['PCC'].min():.6f}; cosine min {g16['cosine'].min():.6f}")
    assert g16["PCC"].min() > 0.9999 and g16["cosine"].min() > 0.9999 and g16["PCC"].min() >= gm["PCC"].min()


def test_config2_shape_depth6_training_iteration_against_oracle(progress):
    """configs[2]'s per-rank step at the FULL model: 512 cells x 2000 genes, hidden 2048, depth 6 (478.6 M parameters,
    1.9 GB of gradients -- the data-parallel bucket payload), fp32: loss, gradient norm, every gradient and the
    parameters after clip + AdamW against the oracle's autograd (about half a minute of host time)."""
    from stdiff_amd.train_engine import TrainEngine
    G, H, depth, heads, C, T = 2000, 2048, 6, 16, 512, 1000
    m, p = build(G, H, depth, heads, 17, "fp32")
    assert sum(v.numel() for v in p.values()) == 3 * G * H + 111 * H * H + 358 * H + G
    mask, _ = problem(8, G)
    st = (O.synth_expression(C, G, 4) * 2 - 1).astype(np.float32)
    gen = torch.Generator().manual_seed(12)
    noise = torch.randn(C, G, generator=gen)
    tt = torch.randint(1, T, (C,), generator=gen)
    x = torch.from_numpy(st)
    xc = torch.from_numpy(st * mask * 1.0).float()
    tb = O.make_tables(T, schedule="cosine")
    progress("configs[2] depth 6: model built, oracle autograd running")
    loss_r, grads, _ = O.loss_and_grads(p, tb, x, xc, noise, tt, mask, heads, depth)
    gn_r = float(sum(g.double().pow(2).sum() for g in grads.values()) ** 0.5)
    progress("configs[2] depth 6: oracle done")
    te = TrainEngine(m, lr=1e-4, diffusion_step=T, mask=mask)
    assert len(te.buckets) == depth + 2 and te.total * 4 > 1.9e9
    loss, gn = te.step(x.to(DEV), xc.to(DEV), noise=noise.to(DEV), timesteps=tt.to(DEV))
    print(f"configs[2] depth 6 fp32: loss {float(loss):.6f} (oracle {float(loss_r):.6f}), |g| {float(gn):.5f} (oracle {gn_r:.5f})")
    assert abs(float(loss) - float(loss_r)) < 1e-5 * max(1.0, float(loss_r))
    assert abs(float(gn) - gn_r) < 1e-4 * gn_r
    gscale = max(float(g.abs().max()) for g in grads.values())
    for n, g in grads.items():
        assert float((te.g[n].cpu() - g).abs().max()) < 1e-4 * gscale, n
