"""GPU: stdiff_gemm_static_scratch -- the one entry point that allocates nothing (split-K partial products in a
file-scope __device__ array of the library, address from hipGetSymbolAddress per device) -- against stdiff_gemm with a
caller workspace of the same size: the same kernels, the same K-split, so the results must be bit-identical
(fp32 and bf16 operands, plain and gated-residual epilogues), on two calls in a row, the second with a smaller input
(the array keeps the first call's partial products).  Collected last on purpose (file name)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
SCRATCH_BYTES = 64 << 20


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


@pytest.mark.parametrize("dt", ["fp32", "bf16"])
def test_static_scratch_gemm_matches_workspace_gemm(dt):
    from stdiff_amd import _lib as L
    from stdiff_amd import ops
    tdt = torch.bfloat16 if dt == "bf16" else torch.float32
    ws = torch.empty(SCRATCH_BYTES // 4, device=DEV, dtype=torch.float32)
    # (M, N, K): batch-sized outputs with a long K (the training GEMMs that are cut along K), then a smaller, ragged one
    for i, (M, N, K) in enumerate([(512, 1024, 4096), (200, 328, 1024)]):
        A = rnd(M, K, seed=10 + i).to(DEV).to(tdt)
        W = rnd(N, K, seed=20 + i, scale=K ** -0.5).to(DEV).to(tdt)
        b = rnd(N, seed=30 + i).to(DEV)
        gate = rnd(M, N, seed=40 + i).to(DEV).to(tdt)
        res = rnd(M, N, seed=50 + i).to(DEV)
        for epi, kw in ((L.EPI_BIAS, {}), (L.EPI_GATE_RES, dict(gate=gate, res=res))):
            ref = torch.empty(M, N, device=DEV, dtype=torch.float32)
            got = torch.full((M, N), float("nan"), device=DEV, dtype=torch.float32)
            ops.gemm(A, W, b, out=ref, epi=epi, workspace=ws, **kw)
            ops.gemm(A, W, b, out=got, epi=epi, static_scratch=True, **kw)
            torch.cuda.synchronize()
            assert torch.equal(ref, got), (dt, M, N, K, epi)
            # and against a plain fp32 product (the split really sums the slices)
            full = A.float() @ W.float().t() + b
            if epi == L.EPI_GATE_RES:
                full = res + gate.float() * full
            assert float((got - full).abs().max()) < (2e-2 if dt == "bf16" else 1e-4) * float(full.abs().max())


def test_static_scratch_rejects_oversized_splits_only(gemm_knobs):
    """Shapes that never split along K need no scratch and pass through whatever their size (round-2 advice: the
    131072-row GEMMs of the sampler were rejected by a blanket size check); a shape that WOULD split but whose smallest
    split (3 slices of the fp32 partial product) does not fit the 64 MiB array is rejected before anything is launched.
    With the default knobs a splitting shape has at most CUs/3 tiles of 128 x 128 (17 MB for 3 slices), so the limit is
    only reachable with a raised STDIFF_SPLITK_BLOCKS."""
    from stdiff_amd import ops
    M, N, K = 4096, 4096, 1024                    # one slice of the partial product = 64 MiB
    A = torch.randn(M, K, device=DEV)
    W = torch.randn(N, K, device=DEV) * K ** -0.5
    ref = ops.gemm(A, W, None, out=torch.empty(M, N, device=DEV))
    out = torch.zeros(M, N, device=DEV)
    ops.gemm(A, W, None, out=out, static_scratch=True)        # 1024 tiles: no split, no scratch, no size limit
    assert torch.equal(out, ref)
    gemm_knobs(STDIFF_SPLITK_BLOCKS="100000")                 # now the host WANTS 16 slices of 64 MiB each
    out.zero_()
    with pytest.raises(ValueError, match="static scratch"):
        ops.gemm(A, W, None, out=out, static_scratch=True)
    assert float(out.abs().max()) == 0.0            # nothing was launched
