"""GPU: BASELINE.json configs[3] -- 1 000 000 spots x 1024 genes, hidden 1024, chunks of 512 (1954 chunks, the last one
64 rows) -- the configuration the headline metric is quoted on, at its FULL size on one GPU.

The oracle cannot run a million rows, so the full run is checked through size-independent properties, and the oracle
evaluates single chunks picked from it (first chunk, the chunks either side of a row-slab boundary, the ragged last):
  * known genes stay bit-equal to gt after every step (mask splice, sample.py:93-94),
  * a chunk's result depends on nothing but the chunk (attention is per chunk, noise is keyed by the global row):
    chunk k run alone == chunk k inside the full 2-lane, 8-slab run, bit for bit,
  * teacher-forced denoiser output of those chunks against the oracle: 1e-5 (fp32 path) / 1e-3 (fp16 operands) of the
    output scale (max-norm relative error: max|a-b| / max|b|); next to it the element-wise form of the same tolerance
    is reported and asserted (conftest.elementwise: fraction of elements within rtol|b| + atol).
Inputs are drawn on the device (a 4 GB host matrix would take minutes to synthesise); only the checked chunks travel
to the host."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import cpu_ref as O  # noqa: E402

DEV = "cuda:0"
N, G, H, DEPTH, HEADS, B, T = 1_000_000, 1024, 1024, 6, 16, 512, 1000
CHUNKS = (0, 255, 256, 977, 1953)          # slab = 256 chunks; lane 1 starts at chunk 977; 1953 = the 64-row chunk


def relerr(a, b):
    a, b = torch.as_tensor(a).float().cpu(), torch.as_tensor(b).float().cpu()
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30)


def setup(dtype):
    from stdiff_amd.model.stDiff_model import DiT_stDiff
    from stdiff_amd.synth import synth_expression_device
    m = DiT_stDiff(G, H, DEPTH, 'dit', HEADS, 6, compute_dtype=dtype)
    p = O.make_params(G, H, DEPTH, 31)
    m.load_state_dict(p, strict=True)
    m = m.to(DEV).eval()
    mask = O.kfold_mask(G, 0)
    md = torch.from_numpy(mask).to(DEV)
    gt = synth_expression_device(N, G, 5, DEV)
    gt.mul_(md).mul_(2).sub_(1)
    lens = [b - a for a, b in O.chunk_bounds(N, B)]
    assert len(lens) == 1954 and lens[-1] == 64
    return m, p, mask, md, gt, lens


def test_config3_full_size_properties_fp16():
    from stdiff_amd import engine as E
    from stdiff_amd.model.stDiff_scheduler import NoiseScheduler
    m, p, mask, md, gt, lens = setup("fp16")
    ns = NoiseScheduler(T, beta_schedule="cosine")
    ts = list(range(T))[::-1][:3]
    known = torch.from_numpy(mask.astype(bool)).to(DEV)
    with torch.no_grad():
        full = E.LanedSampler(m.engine(), ns, lens, gt, gt, md, seed=1234, lanes=2)
        assert len(full.parts) == 2 and len(full.parts[0].slabs) >= 4           # row slabs share one workspace
        full.init()
        out = full.run(ts)
        torch.cuda.synchronize()
        for lo in range(0, N, 1 << 17):                                          # slab by slab: no 4 GB temporaries
            o, g = out[lo:lo + (1 << 17)], gt[lo:lo + (1 << 17)]
            assert bool(torch.isfinite(o).all())
            assert torch.equal(o[:, known], g[:, known])                         # splice: known genes bit-equal to gt
        assert float((out[:4096, ~known] - gt[:4096, ~known]).abs().max()) > 0   # unknown genes are being sampled
        for k in CHUNKS:                                                         # chunk alone == chunk in the full run
            lo = k * B
            hi = lo + lens[k]
            s = E.ReverseSampler(m.engine(), ns, [lens[k]], gt[lo:hi], gt[lo:hi], md, seed=1234, row_offset=lo, lane=7)
            s.init()
            assert torch.equal(s.run(ts), out[lo:hi]), k


@pytest.mark.parametrize("dtype", ["fp32", "fp16"])
def test_config3_chunks_against_oracle(dtype, progress):
    """Teacher-forced denoiser call over the full 1 M-row matrix; the oracle evaluates five of its 1954 chunks, and
    the fused scheduler step of those chunks is checked bit-for-bit-tolerance against the oracle's with injected z."""
    from stdiff_amd import engine as E
    from stdiff_amd.model.stDiff_scheduler import NoiseScheduler
    m, p, mask, md, gt, lens = setup(dtype)
    t = 431
    gen = torch.Generator(device=DEV).manual_seed(6)
    mk = torch.from_numpy(mask)
    tol = 1e-5 if dtype == "fp32" else 1e-3
    tb = O.make_tables(T, schedule="cosine")
    with torch.no_grad():
        smp = E.ReverseSampler(m.engine(), NoiseScheduler(T, beta_schedule="cosine"), lens, gt, gt, md, use_graph=False)
        for lo in range(0, N, 1 << 17):                                          # x_t = randn spliced with gt
            x = torch.randn((min(1 << 17, N - lo), G), device=DEV, generator=gen)
            smp.x[lo:lo + x.shape[0]].copy_(x * (1 - md) + gt[lo:lo + x.shape[0]] * md)
        smp.set_x(smp.x.clone())
        eps = smp.denoise_only(t)
        worst = 0.0
        for k in CHUNKS:
            lo = k * B
            hi = lo + lens[k]
            x_c, gt_c = smp.x[lo:hi].cpu(), gt[lo:hi].cpu()
            ref = O.dit_forward(p, x_c, torch.full((hi - lo,), t, dtype=torch.int64), gt_c, HEADS, DEPTH)
            e = relerr(eps[lo:hi], ref)
            worst = max(worst, e)
            progress(f"configs[3] {dtype} chunk {k} ({hi - lo} rows): eps rel err {e:.3e}")
            assert e < tol, k
            from conftest import elementwise_line
            ew, line = elementwise_line(f"configs[3] {dtype} chunk {k}", eps[lo:hi].cpu().numpy(), ref.numpy(), tol)
            progress(line)
            assert ew["full"] == 1.0 and ew["tenth"] > 0.8, (k, ew)
            if dtype == "fp32" and k in (0, 1953):
                z = torch.randn(hi - lo, G, generator=torch.Generator().manual_seed(k))
                ref_x, _ = O.step(tb, ref, t, x_c, z)
                ref_x = ref_x * (1 - mk) + gt_c * mk
                s = E.ReverseSampler(m.engine(), NoiseScheduler(T, beta_schedule="cosine"), [hi - lo], gt[lo:hi], gt[lo:hi],
                                     md, use_graph=False, lane=7)
                s.set_x(smp.x[lo:hi].clone())
                got = s.run([t], z=z.to(DEV)[None]).cpu()
                assert relerr(got, ref_x) < tol, k
