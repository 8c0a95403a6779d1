"""GPU: every HIP entry point against a plain fp32 statement of the same op (torch on the CPU), through the C ABI.
Tolerances: fp32 path 1e-5 of the output scale (north_star); bf16 path compared with the SAME op evaluated in fp32
on bf16-rounded operands (so only accumulation order and the output rounding differ)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import cpu_ref as O  # noqa: E402


@pytest.fixture(scope="module")
def ops():
    from stdiff_amd import ops as _ops
    return _ops


@pytest.fixture(scope="module")
def L():
    from stdiff_amd import _lib
    return _lib


DEV = "cuda:0"


def relerr(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30)


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


TDT = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}


def to_dt(x, dt):
    return x.to(DEV).to(TDT[dt]).contiguous()


def q(x, dt):
    """operand as the kernel sees it, back in fp32"""
    return x.to(TDT[dt]).float()


# ---------------------------------------------------------------- GEMM
@pytest.mark.parametrize("dt", ["fp32", "bf16", "fp16"])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 136, 128), (77, 1000, 192), (512, 3072, 1024), (1, 8, 64),
                                   (300, 33, 64)])
def test_gemm_bias(ops, L, dt, M, N, K):
    A, W, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3)
    ref = q(A, dt) @ q(W, dt).T + b
    out = ops.gemm(to_dt(A, dt), to_dt(W, dt), b.to(DEV), out_dtype=torch.float32)
    assert relerr(out, ref) < (2e-6 if dt == "fp32" else 2e-6 * 8)
    if dt != "fp32":                 # 16-bit output: one rounding of the result (2^-9 bf16, 2^-12 fp16)
        out16 = ops.gemm(to_dt(A, dt), to_dt(W, dt), b.to(DEV))
        assert out16.dtype == TDT[dt] and relerr(out16, ref) < (5e-3 if dt == "bf16" else 6e-4)


def test_fp16_conversions_saturate(ops, L):
    """The fp16 build never writes inf: results beyond +-65504 are clamped (GEMM epilogues, LN-modulate, pack_rows)."""
    M, N, K = 256, 256, 64
    A = torch.full((M, K), 2000.0)
    W = torch.ones(N, K)
    W[N // 2:] = -1.0
    out = ops.gemm(to_dt(A, "fp16"), to_dt(W, "fp16"), None)                 # |v| = 128000 > 65504
    assert out.dtype == torch.float16 and bool(torch.isfinite(out).all())
    assert float(out[:, :N // 2].min()) == 65504.0 and float(out[:, N // 2:].max()) == -65504.0
    exact = ops.gemm(to_dt(A, "fp16"), to_dt(W, "fp16"), None, out_dtype=torch.float32)
    assert float(exact.abs().min()) == 128000.0                                # the fp32 accumulators are exact
    big = torch.tensor([[1e6, -1e6, 3.0, 70000.0] * 16], device=DEV)
    p = ops.pack_rows(big, 64, torch.float16)
    assert p[0, :4].tolist() == [65504.0, -65504.0, 3.0, 65504.0]


def test_lane_streams_are_concurrent(ops):
    """engine.concurrent_streams: distinct streams whose spin probes overlap (one period, not two)."""
    import time
    from stdiff_amd import engine as E
    a, b = E.concurrent_streams(torch.device(DEV), 2)
    assert a.cuda_stream != b.cuda_stream
    ops.spin_probe(1, 1)
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for st in (a, b):
            with torch.cuda.stream(st):
                ops.spin_probe(64, 2000)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    assert 1.9e-3 < best < 3.2e-3, best
    with pytest.raises(ValueError):
        ops.spin_probe(0, 10)


def test_gemm_layout_exact(ops):
    """A = I against an ASYMMETRIC integer W: catches any row/col or k-permutation slip exactly."""
    for dt, K in (("bf16", 128), ("fp32", 64)):
        M = N = 256
        A = torch.zeros(M, K)
        A[torch.arange(K), torch.arange(K)] = 1.0
        W = (torch.arange(N)[:, None] * 3 + torch.arange(K)[None, :] * 7) % 61 - 30.0
        out = ops.gemm(to_dt(A, dt), to_dt(W, dt), None, out_dtype=torch.float32)
        assert torch.equal(out.cpu(), A @ W.T), dt


@pytest.mark.parametrize("dt", ["fp32", "bf16"])
def test_gemm_epilogues(ops, L, dt):
    M, N, K = 150, 264, 128
    A, W, b = rnd(M, K, seed=4), rnd(N, K, seed=5, scale=K ** -0.5), rnd(N, seed=6, scale=0.1)
    v = q(A, dt) @ q(W, dt).T + b
    Ad, Wd, bd = to_dt(A, dt), to_dt(W, dt), b.to(DEV)
    tol = 3e-6 if dt == "fp32" else 3e-5
    assert relerr(ops.gemm(Ad, Wd, bd, out_dtype=torch.float32, epi=L.EPI_SILU), F.silu(v)) < tol
    assert relerr(ops.gemm(Ad, Wd, bd, out_dtype=torch.float32, epi=L.EPI_GELU), F.gelu(v)) < tol
    res, gate = rnd(M, N, seed=7), rnd(M, 3 * N, seed=8)
    gd = to_dt(gate, dt)
    h = res.to(DEV).clone()
    ops.gemm(Ad, Wd, bd, out=h, epi=L.EPI_GATE_RES, gate=gd[:, N:2 * N], res=h)       # in place on the residual
    assert relerr(h, res + q(gate, dt)[:, N:2 * N] * v) < tol
    assert relerr(ops.gemm(Ad, Wd, bd, out_dtype=torch.float32, epi=L.EPI_ADD, res=res.to(DEV)), v + res) < tol
    acc = res.to(DEV).clone()
    ops.gemm(Ad, Wd, bd, out=acc, epi=L.EPI_ACCUM)
    assert relerr(acc, res + v) < tol
    o1 = torch.empty(M, N, device=DEV, dtype=Ad.dtype)
    o2 = torch.empty(M, N, device=DEV, dtype=Ad.dtype)
    ops.gemm(Ad, Wd, bd, out=o1, epi=L.EPI_GELU_SAVE, out2=o2)
    t2 = 5e-3 if dt == "bf16" else tol
    assert relerr(o1, F.gelu(v)) < t2 and relerr(o2, v) < t2


@pytest.mark.parametrize("dt", ["fp32", "bf16"])
@pytest.mark.parametrize("M,N,K", [(512, 1024, 4096), (200, 136, 1024), (64, 1000, 2048), (300, 33, 512)])
def test_gemm_split_k(ops, L, dt, M, N, K):
    """Few tiles x long K with a caller workspace: deterministic split-K + reduce kernel, every epilogue kind."""
    A, W, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5), rnd(N, seed=3, scale=0.1)
    v = q(A, dt) @ q(W, dt).T + b
    Ad, Wd, bd = to_dt(A, dt), to_dt(W, dt), b.to(DEV)
    ws = torch.empty(8 << 20, device=DEV, dtype=torch.float32)
    tol = 3e-6 if dt == "fp32" else 3e-5
    out = ops.gemm(Ad, Wd, bd, out_dtype=torch.float32, workspace=ws)
    assert relerr(out, v) < tol
    assert torch.equal(out, ops.gemm(Ad, Wd, bd, out_dtype=torch.float32, workspace=ws))          # run-to-run identical
    assert relerr(out, ops.gemm(Ad, Wd, bd, out_dtype=torch.float32)) < tol                       # vs the unsplit kernel
    assert relerr(ops.gemm(Ad, Wd, bd, out_dtype=torch.float32, epi=L.EPI_GELU, workspace=ws), F.gelu(v)) < tol
    res, gate = rnd(M, N, seed=7), rnd(M, N, seed=8)
    h = res.to(DEV).clone()
    ops.gemm(Ad, Wd, bd, out=h, epi=L.EPI_GATE_RES, gate=to_dt(gate, dt), res=h, workspace=ws)
    assert relerr(h, res + q(gate, dt) * v) < tol
    acc = res.to(DEV).clone()
    ops.gemm(Ad, Wd, None, out=acc, epi=L.EPI_ACCUM, workspace=ws)
    assert relerr(acc, res + v - b) < tol
    o1 = torch.empty(M, N, device=DEV, dtype=Ad.dtype)
    o2 = torch.empty(M, N, device=DEV, dtype=Ad.dtype)
    ops.gemm(Ad, Wd, bd, out=o1, epi=L.EPI_GELU_SAVE, out2=o2, workspace=ws)
    t2 = 5e-3 if dt == "bf16" else tol
    assert relerr(o1, F.gelu(v)) < t2 and relerr(o2, v) < t2
    small = torch.empty(16, device=DEV, dtype=torch.float32)                                      # too small: falls back
    assert relerr(ops.gemm(Ad, Wd, bd, out_dtype=torch.float32, workspace=small), v) < tol


@pytest.mark.parametrize("dt", ["fp32", "bf16"])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (200, 136, 40), (1000, 264, 512), (72, 1024, 300), (1024, 1024, 8)])
def test_gemm_k_major_operands(ops, L, dt, M, N, K):
    """Backward-pass layouts without transposed copies: A given as [K, M] and / or W as [K, N]; K (the batch) need not
    be a multiple of the K-slab (rows >= k_valid count as zero).  Against the same contraction in fp32."""
    A, W, b = rnd(M, K, seed=21), rnd(N, K, seed=22, scale=K ** -0.5), rnd(N, seed=23, scale=0.1)
    ref = q(A, dt) @ q(W, dt).T
    Ad, Wd = to_dt(A, dt), to_dt(W, dt)
    At, Wt = Ad.t().contiguous(), Wd.t().contiguous()                 # [K, M], [K, N]
    tol = 3e-6 if dt == "fp32" else 3e-5
    bk = 64 if dt == "bf16" else 16
    Kp = (K + bk - 1) // bk * bk
    pad = lambda x: torch.nn.functional.pad(x, (0, Kp - K)).contiguous()      # row-major operands carry the padded K
    ws = torch.empty(4 << 20, device=DEV, dtype=torch.float32)
    out = ops.gemm(At, Wt, None, out_dtype=torch.float32, a_kmajor=True, w_kmajor=True)            # dW = dY^T X
    assert relerr(out, ref) < tol
    assert relerr(ops.gemm(At, Wt, b.to(DEV), out_dtype=torch.float32, a_kmajor=True, w_kmajor=True, workspace=ws),
                  ref + b) < tol
    assert relerr(ops.gemm(pad(Ad), Wt, None, out_dtype=torch.float32, w_kmajor=True), ref) < tol   # dX = dY W
    assert relerr(ops.gemm(At, pad(Wd), None, out_dtype=torch.float32, a_kmajor=True), ref) < tol
    acc = rnd(M, N, seed=24).to(DEV)
    base = acc.clone()
    ops.gemm(pad(Ad), Wt, None, out=acc, epi=L.EPI_ACCUM, w_kmajor=True, workspace=ws)
    assert relerr(acc, base.cpu() + ref) < tol
    sub = ops.gemm(At, Wt, None, out_dtype=torch.float32, a_kmajor=True, w_kmajor=True, M=M - 8 if M > 8 else M)
    assert relerr(sub, ref[:sub.shape[0]]) < tol                      # a row subset of a wider K-major operand


@pytest.mark.parametrize("dt", ["fp32", "bf16"])
@pytest.mark.parametrize("C", [40, 512, 1000])
def test_dw_group_against_fp32(ops, L, dt, C):
    """stdiff_dw_group: several weight gradients gw_i = dY_i^T X_i and bias gradients gb_i = colsum(dY_i) over one batch
    in one launch (ragged tile counts, a row subset of a wider dY, a strided X, one problem without a bias gradient),
    against the same contractions in fp32 -- and against the one-job path (stdiff_gemm K-major + stdiff_colsum)."""
    shapes = [(768, 128, 768, 128), (136, 264, 200, 264), (128, 1024, 128, 1024), (8, 8, 8, 8), (264, 72, 264, 80)]
    prob, refs = [], []
    for i, (M, N, wy, wx) in enumerate(shapes):
        dY, X = rnd(C, wy, seed=40 + i), rnd(C, wx, seed=50 + i, scale=C ** -0.5)
        gw = torch.full((M, N), 7.0, device=DEV)
        gb = None if i == 3 else torch.full((M,), 7.0, device=DEV)
        prob.append((to_dt(dY, dt), to_dt(X, dt), gw, gb, M))
        refs.append((q(dY, dt)[:, :M].T @ q(X, dt)[:, :N], q(dY, dt)[:, :M].sum(0)))
    ops.dw_group(prob)
    tol = 3e-6 if dt == "fp32" else 3e-5
    total = 0.0
    for (dY, X, gw, gb, M), (rw, rb) in zip(prob, refs):
        assert relerr(gw, rw) < tol
        one = ops.gemm(dY, X[:, :gw.shape[1]], None, out_dtype=torch.float32, a_kmajor=True, w_kmajor=True, M=M)
        assert relerr(gw, one) < tol
        total += float(gw.double().pow(2).sum())
        if gb is not None:
            assert relerr(gb, rb) < tol
            total += float(gb.double().pow(2).sum())
    # the launch's share of the squared gradient norm (args.sq_parts): per-wave sums of squares of everything it wrote,
    # ragged edges excluded, every slot overwritten; stdiff_sum_parts adds them in a fixed order
    kept = [(gw.clone(), None if gb is None else gb.clone()) for _, _, gw, gb, _ in prob]
    n = ops.dw_group_parts(prob)
    parts = torch.full((n + 8,), float("nan"), device=DEV)
    ops.dw_group(prob, sq_parts=parts[:n])
    assert bool(torch.isfinite(parts[:n]).all()) and bool(torch.isnan(parts[n:]).all())
    ss = torch.full((1,), 3.0, device=DEV)
    ops.sum_parts(parts[:n], ss)
    assert abs(float(ss) - 3.0 - total) < 2e-6 * total
    again = torch.empty(1, device=DEV)
    ops.sum_parts(parts[:n], again, accumulate=False)
    assert abs(float(again) - total) < 2e-6 * total and abs(float(again) - float(parts[:n].double().sum())) < 1e-6 * total
    for (_, _, gw, gb, _), (kw, kb) in zip(prob, kept):
        assert torch.equal(gw, kw) and (gb is None or torch.equal(gb, kb))
    # no cells: the parts are still all written (zeros); an empty parts list leaves / clears the cell
    parts.fill_(float("nan"))
    ops.dw_group(prob, C_rows=0, sq_parts=parts[:n])
    assert float(parts[:n].abs().sum()) == 0.0 and bool(torch.isnan(parts[n:]).all())
    ops.sum_parts(parts[:0], ss)
    assert abs(float(ss) - 3.0 - total) < 2e-6 * total
    ops.sum_parts(parts[:0], ss, accumulate=False)
    assert float(ss) == 0.0
    with pytest.raises((ValueError, RuntimeError)):          # 16-byte column pieces do not exist for 33 columns
        ops.dw_group([(prob[0][0], prob[0][1], torch.zeros(33, 128, device=DEV), None, 33)])


@pytest.mark.parametrize("variant", ["3", "4", "4s", "4h"])
@pytest.mark.parametrize("M,N,K", [(600, 520, 256), (257, 264, 64), (1024, 512, 1024), (300, 1000, 192)])
def test_gemm_256_kernel_all_epilogues(ops, L, M, N, K, variant, gemm_knobs):
    """The persistent 256 x 256 bf16 kernel forced (STDIFF_GEMM_TILE=3 through the gemm_knobs fixture): ragged M / N edges, every
    epilogue kind, fp32 and bf16 outputs, K of 1 / 3 / 4 / 16 slabs (prologue and tail of the refill schedule).  Variant 3 =
    four 16-MFMA phases per slab, 4 = two 32-MFMA phases (the default), 4s = the same with the tiles of a short last
    round cut into 128-row half units (STDIFF_GEMM_SPLIT_TAIL=1; here every tile is in that round), 4h = the default
    kernel of the fp16 build (libstdiff_hip_f16.so)."""
    dt = "fp16" if variant.endswith("h") else "bf16"
    L.lib(dt == "fp16")                                   # loaded before the knobs are published to it
    gemm_knobs(STDIFF_GEMM_TILE=variant[0], **({"STDIFF_GEMM_SPLIT_TAIL": "1"} if variant.endswith("s") else {}))
    dt = "fp16" if variant.endswith("h") else "bf16"
    t16 = 6e-4 if dt == "fp16" else 5e-3          # one rounding of a 16-bit result
    A, W, b = rnd(M, K, seed=11), rnd(N, K, seed=12, scale=K ** -0.5), rnd(N, seed=13, scale=0.1)
    v = q(A, dt) @ q(W, dt).T + b
    Ad, Wd, bd = to_dt(A, dt), to_dt(W, dt), b.to(DEV)
    tol = 3e-5
    assert relerr(ops.gemm(Ad, Wd, bd, out_dtype=torch.float32), v) < tol
    assert relerr(ops.gemm(Ad, Wd, None, out_dtype=torch.float32), v - b) < tol
    assert relerr(ops.gemm(Ad, Wd, bd), v) < t16
    assert relerr(ops.gemm(Ad, Wd, bd, out_dtype=torch.float32, epi=L.EPI_SILU), F.silu(v)) < tol
    assert relerr(ops.gemm(Ad, Wd, bd, out_dtype=torch.float32, epi=L.EPI_GELU), F.gelu(v)) < tol
    g16 = ops.gemm(Ad, Wd, bd, epi=L.EPI_GELU)                      # bf16 output: the polynomial GELU
    assert relerr(g16, F.gelu(v)) < t16
    res, gate = rnd(M, N, seed=7), rnd(M, 3 * N, seed=8)
    gd = to_dt(gate, dt)
    h = res.to(DEV).clone()
    ops.gemm(Ad, Wd, bd, out=h, epi=L.EPI_GATE_RES, gate=gd[:, N:2 * N], res=h)
    assert relerr(h, res + q(gate, dt)[:, N:2 * N] * v) < tol
    assert relerr(ops.gemm(Ad, Wd, bd, out_dtype=torch.float32, epi=L.EPI_ADD, res=res.to(DEV)), v + res) < tol
    acc = res.to(DEV).clone()
    ops.gemm(Ad, Wd, bd, out=acc, epi=L.EPI_ACCUM)
    assert relerr(acc, res + v) < tol
    o1 = torch.empty(M, N, device=DEV, dtype=Ad.dtype)
    o2 = torch.empty(M, N, device=DEV, dtype=Ad.dtype)
    ops.gemm(Ad, Wd, bd, out=o1, epi=L.EPI_GELU_SAVE, out2=o2)
    assert relerr(o1, F.gelu(v)) < t16 and relerr(o2, v) < t16


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
def test_gemm_128_dma_variant_equals_register_staged(ops, L, gemm_knobs, dt):
    """gemm128d_kernel (the 128 x 128 tile with LDS-DMA staging and eight waves, STDIFF_GEMM_DMA128) against gemm_nt_kernel
    (register staging, four waves): the same fragments into the same MFMAs in the same order, so the same bits -- ragged M / N edges, W
    row-major and K-major (with a K that is not a multiple of the slab: k_valid < K), 1 ... 64 K-slabs (prologue, steady
    state and tail of the four-buffer ring), split-K through a workspace, and every epilogue kind."""
    L.lib(dt == "fp16")
    ws = torch.empty(64 << 20, device=DEV, dtype=torch.uint8)

    def both(fn):
        outs = []
        for dma in ("0", "2"):                 # 2: K-major W as well (the default, 1, covers row-major W only)
            gemm_knobs(STDIFF_GEMM_TILE="1", STDIFF_GEMM_DEPTH="3", STDIFF_GEMM_DMA128=dma)
            outs.append(fn())
        for x, y in zip(*outs):
            assert torch.equal(x.view(torch.int32 if x.dtype == torch.float32 else torch.int16),
                               y.view(torch.int32 if y.dtype == torch.float32 else torch.int16))
        return outs[0]

    for (M, N, K) in [(512, 1024, 1024), (96, 72, 64), (300, 520, 192), (512, 128, 4096), (130, 1000, 320)]:
        A, W, b = to_dt(rnd(M, K, seed=51), dt), to_dt(rnd(N, K, seed=52, scale=K ** -0.5), dt), rnd(N, seed=53).to(DEV)
        ref = A.float() @ W.float().T + b
        got = both(lambda: [ops.gemm(A, W, b, out_dtype=torch.float32, workspace=ws), ops.gemm(A, W, b, workspace=ws),
                            ops.gemm(A, W, None, out_dtype=torch.float32, epi=L.EPI_SILU),
                            ops.gemm(A, W, b, epi=L.EPI_GELU, workspace=ws)])
        assert relerr(got[0], ref) < 3e-5
        res, gate = rnd(M, N, seed=54).to(DEV), to_dt(rnd(M, 2 * N, seed=55), dt)

        def side():
            h = torch.full((M, N), float("nan"), device=DEV)
            v = torch.full((M, N), float("nan"), device=DEV, dtype=A.dtype)
            ops.gemm(A, W, b, out=h, epi=L.EPI_GATE_RES, gate=gate[:, N:], res=res, out2=v, workspace=ws)
            acc = res.clone()
            ops.gemm(A, W, b, out=acc, epi=L.EPI_ACCUM, workspace=ws)
            o1 = torch.empty(M, N, device=DEV, dtype=A.dtype)
            o2 = torch.empty(M, N, device=DEV, dtype=A.dtype)
            ops.gemm(A, W, b, out=o1, epi=L.EPI_GELU_SAVE, out2=o2)
            return [h, v, acc, ops.gemm(A, W, b, out_dtype=torch.float32, epi=L.EPI_ADD, res=res), o1, o2]
        both(side)
    # W K-major: dX = dY W with W [out = K, in = N]; K = 200 output features (k_valid 200 of 256), and a GELU-backward epilogue
    for (M, Kv, N) in [(512, 1024, 1024), (100, 200, 136), (512, 4096, 1024), (64, 72, 2048)]:
        dY, Wk = to_dt(rnd(M, (Kv + 63) // 64 * 64, seed=61), dt), to_dt(rnd(Kv, N, seed=62, scale=Kv ** -0.5), dt)
        dY[:, Kv:] = 0
        upre = to_dt(rnd(M, N, seed=63), dt)
        got = both(lambda: [ops.gemm(dY, Wk, None, out_dtype=torch.float32, w_kmajor=True, workspace=ws),
                            ops.gemm(dY, Wk, None, w_kmajor=True, workspace=ws),
                            ops.gemm(dY, Wk, None, w_kmajor=True, epi=L.EPI_GELU_BWD, gate=upre)])
        assert relerr(got[0], dY[:, :Kv].float() @ Wk.float()) < 3e-5


@pytest.mark.parametrize("dt", ["bf16", "fp16"])
def test_gemm_192t_equals_256q(ops, L, gemm_knobs, dt):
    """gemm192t_kernel (192 x 256 tile, the previous tile leaves during the next tile's K loop; opt-in with
    STDIFF_GEMM_TILE=5, profiles/notes/r03_gemm192t.md) against gemm256q_kernel: the same bits -- ragged M / N edges, with
    and without bias, bias and GELU epilogues, K of 13 and 64 slabs, and a shape that walks several tiles per CU (the
    pipelined path: operand ring and bias across tile boundaries)."""
    L.lib(dt == "fp16")
    for (M, N, K) in [(192, 256, 832), (1000, 520, 832), (777, 1032, 4096), (30000, 3072, 1024)]:
        A, W, b = to_dt(rnd(M, K, seed=31), dt), to_dt(rnd(N, K, seed=32, scale=K ** -0.5), dt), rnd(N, seed=33).to(DEV)
        for epi in (L.EPI_BIAS, L.EPI_GELU):
            for bias in (b, None):
                gemm_knobs(STDIFF_GEMM_TILE="4")
                ref = ops.gemm(A, W, bias, epi=epi)
                gemm_knobs(STDIFF_GEMM_TILE="5")
                got = torch.full_like(ref, float("nan"))
                ops.gemm(A, W, bias, out=got, epi=epi)
                assert torch.equal(ref.view(torch.int16), got.view(torch.int16)), (M, N, K, epi, bias is not None)


def test_gelu_polynomial_accuracy(ops, L, gemm_knobs):
    """gelu_poly4 (bf16 outputs of the 256 x 256 kernel) against erf GELU over the whole input range: the result,
    read back as bf16, is within one bf16 ulp of the correctly rounded value."""
    gemm_knobs(STDIFF_GEMM_TILE="3")
    M, K = 512, 64
    A = torch.zeros(M, K)
    W = torch.zeros(256, K)
    x_cols = torch.linspace(-9, 9, 256).to(torch.bfloat16).float()
    A[:, 0] = 1.0
    W[:, 0] = x_cols
    out = ops.gemm(to_dt(A, "bf16"), to_dt(W, "bf16"), None, epi=L.EPI_GELU)       # every row = gelu(x_cols)
    ref = F.gelu(x_cols.double()).float()
    got = out[0].float().cpu()
    ulp = torch.maximum(ref.abs(), torch.tensor(1e-30)) * 2.0 ** -8
    assert bool(((got - ref).abs() <= ulp + 2e-6).all())
    assert torch.equal(out[0], out[-1])


def test_gemm_rejects_bad_shapes(ops):
    A, W = torch.zeros(8, 40, device=DEV), torch.zeros(8, 40, device=DEV)
    with pytest.raises(ValueError):
        ops.gemm(A, W)                                   # K % 16 != 0
    with pytest.raises(RuntimeError):
        ops.gemm(torch.zeros(8, 64), torch.zeros(8, 64))  # CPU tensors: no fallback


@pytest.mark.parametrize("dt", ["fp32", "bf16"])
def test_transpose(ops, dt):
    x = to_dt(rnd(130, 77, seed=9), dt)
    assert torch.equal(ops.transpose(x).cpu(), x.cpu().T)


# ---------------------------------------------------------------- attention
def attn_ref(qkv, lens, heads):
    R, H3 = qkv.shape
    H = H3 // 3
    d = H // heads
    outs, lses = [], []
    a = 0
    for n in lens:
        blk = qkv[a:a + n].reshape(n, 3, heads, d).permute(1, 2, 0, 3)
        s = (blk[0] @ blk[1].transpose(-2, -1)) * d ** -0.5
        outs.append((s.softmax(-1) @ blk[2]).transpose(0, 1).reshape(n, H))
        lses.append((torch.logsumexp(s, -1) / np.log(2.0)).T)
        a += n
    return torch.cat(outs), torch.cat(lses)


@pytest.mark.parametrize("dt", ["fp32", "bf16", "fp16"])
@pytest.mark.parametrize("d,heads,lens", [(32, 4, [96]), (32, 16, [40, 40, 16]), (64, 16, [512, 130, 1]),
                                          (128, 16, [200, 64]), (64, 2, [2048, 33])])
def test_attention_fwd(ops, dt, d, heads, lens):
    H = d * heads
    R = sum(lens)
    qkv = rnd(R, 3 * H, seed=d + R)
    ref, lse_ref = attn_ref(q(qkv, dt), lens, heads)
    cu = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int32, device=DEV)
    lse = torch.empty(R, heads, device=DEV)
    out = ops.attention_fwd(to_dt(qkv, dt), cu, max(lens), heads, lse=lse)
    # 16-bit: P and the output are rounded to the operand format
    assert relerr(out, ref) < {"fp32": 1e-5, "bf16": 1.2e-2, "fp16": 1.5e-3}[dt]
    assert relerr(lse, lse_ref) < 1e-5


# ---------------------------------------------------------------- fused bandwidth kernels
@pytest.mark.parametrize("dt", ["fp32", "bf16"])
@pytest.mark.parametrize("R,H", [(37, 128), (100, 512), (65, 1024), (9, 2048)])
def test_ln_modulate(ops, dt, R, H):
    x, mod = rnd(R, H, seed=1) * 2 + 0.3, rnd(R, 6 * H, seed=2, scale=0.5)
    md = to_dt(mod, dt)
    ref = F.layer_norm(x, (H,), eps=1e-6) * (1 + q(mod, dt)[:, 4 * H:5 * H]) + q(mod, dt)[:, 3 * H:4 * H]
    mean, rstd = torch.empty(R, device=DEV), torch.empty(R, device=DEV)
    out = ops.ln_modulate(x.to(DEV), md[:, 3 * H:4 * H], md[:, 4 * H:5 * H], mean=mean, rstd=rstd)
    assert relerr(out, ref) < (2e-6 if dt == "fp32" else 5e-3)
    assert relerr(mean, x.mean(1)) < 1e-5 and relerr(rstd, (x.var(1, unbiased=False) + 1e-6).rsqrt()) < 1e-5


def test_silu_cond_and_embedding(ops):
    from stdiff_amd.engine import sinusoid_freqs
    R, H, T = 50, 128, 30
    temb, y = rnd(T, H, seed=1), rnd(R, H, seed=2)
    tix = torch.tensor([17], dtype=torch.int32, device=DEV)
    out = ops.silu_cond(y.to(DEV), temb=temb.to(DEV), t_index=tix)
    assert relerr(out, F.silu(temb[17] + y)) < 2e-6
    assert relerr(ops.silu_cond(y.to(DEV)), F.silu(y)) < 2e-6
    t = torch.tensor([0, 1, 2, 499, 998, 999, 1499], dtype=torch.int64)
    e = ops.timestep_embedding(t.to(DEV), sinusoid_freqs(DEV))
    assert relerr(e, O.timestep_embedding(t)) < 2e-6
    x = rnd(11, 33, seed=3)
    p = ops.pack_rows(x.to(DEV), 64, torch.bfloat16)
    assert torch.equal(p[:, :33].cpu(), x.to(torch.bfloat16)) and float(p[:, 33:].abs().max()) == 0.0


# ---------------------------------------------------------------- diffusion arithmetic: bit-exact vs the reference
def test_ddpm_step_bit_exact(ops, golden_dir):
    import os
    from stdiff_amd.model.stDiff_scheduler import NoiseScheduler
    from stdiff_amd.engine import step_coefficients
    g = np.load(os.path.join(golden_dir, "g4_step.npz"))
    ns = NoiseScheduler(1000, beta_schedule="cosine")
    # torch's CPU sqrt/div (MKL VML) is not correctly rounded and differs by 1 ulp between host CPUs, so the
    # reference's own tables are host dependent: bit-exactness is claimed GIVEN the tables the reference had
    # (tests/golden/g3_tables.npz, recorded next to g4 on the same host).
    g3 = np.load(os.path.join(golden_dir, "g3_tables.npz"))
    for n in O.TABLE_NAMES:
        setattr(ns, n, torch.from_numpy(g3[f"cosine_1000_{n}"]))
    coef = step_coefficients(ns)
    coef[1:, 4] = torch.from_numpy(np.sqrt(g3["cosine_1000_variance"].astype(np.float32))[1:])   # 0-dim ** 0.5
    ns._dev.clear()
    ns._device_tables(torch.device(DEV))["coef"] = coef.to(DEV)
    coef = coef.to(DEV)
    tdev = torch.zeros(1, dtype=torch.int32, device=DEV)
    for t in (999, 500, 1, 0):
        G = g[f"t{t}_x"].shape[1]
        zero, m0 = torch.zeros_like(torch.from_numpy(g[f"t{t}_x"])).to(DEV), torch.zeros(G, device=DEV)
        for mode, key in ((0, "prev"), (1, "prev_xstart")):
            x = torch.from_numpy(g[f"t{t}_x"]).to(DEV)
            tdev.fill_(t)
            ops.ddpm_step(x, torch.from_numpy(g[f"t{t}_eps"]).to(DEV), zero, m0, coef, tdev, t,
                          z_base=torch.from_numpy(g[f"t{t}_z"]).to(DEV), pred_type=mode)
            assert np.array_equal(x.cpu().numpy(), g[f"t{t}_{key}"]), (t, mode)
        # the scheduler object (reference API) returns the same pair
        prev, x0 = ns.step(torch.from_numpy(g[f"t{t}_eps"]).to(DEV), torch.tensor(t), torch.from_numpy(g[f"t{t}_x"]).to(DEV),
                           noise=torch.from_numpy(g[f"t{t}_z"]).to(DEV))
        assert np.array_equal(prev.cpu().numpy(), g[f"t{t}_prev"]) and np.array_equal(x0.cpu().numpy(), g[f"t{t}_x0"])
    out = ns.add_noise(torch.from_numpy(g["an_x0"]).to(DEV), torch.from_numpy(g["an_noise"]).to(DEV),
                       torch.from_numpy(g["an_t"]))
    assert np.array_equal(out.cpu().numpy(), g["an_out"])
    with pytest.raises(NotImplementedError):
        ns.step(out, torch.tensor(3), out, model_pred_type="v")


def test_scheduler_api_completeness(ops):
    """S3/S4/S7 of SURVEY 8(a): reconstruct_x0, q_posterior, undo and the linear / quadratic schedules through the
    reference's method names, against the oracle's restatement."""
    from stdiff_amd.model.stDiff_scheduler import NoiseScheduler
    for sched in ("cosine", "linear", "quadratic"):
        T = 60
        ns = NoiseScheduler(T, beta_schedule=sched)
        tb = O.make_tables(T, schedule=sched)
        x, e = rnd(21, 36, seed=1), rnd(21, 36, seed=2)
        for t in (T - 1, 7, 0):
            x0 = ns.reconstruct_x0(x.to(DEV), t, e.to(DEV))
            assert np.array_equal(x0.cpu().numpy(), O.reconstruct_x0(tb, x, t, e).numpy()), (sched, t)
            mean = ns.q_posterior(x0, x.to(DEV), t)
            assert np.array_equal(mean.cpu().numpy(), O.q_posterior(tb, x0.cpu(), x, t).numpy()), (sched, t)
            v = ns.get_variance(torch.tensor(t))
            assert float(v) == float(O.variance(tb, t))
        torch.manual_seed(11)
        und = ns.undo(None, x.to(DEV), None, 5)
        torch.manual_seed(11)
        z = torch.randn_like(x.to(DEV)).cpu()
        beta = tb["betas"][5]
        assert relerr(und, torch.sqrt(1 - beta) * x + torch.sqrt(beta) * z) < 1e-6
        assert len(ns) == T


def test_step_splice_and_ragged_width(ops):
    """G not a multiple of 4 (scalar path) + mask splice + packed operand refresh."""
    from stdiff_amd.model.stDiff_scheduler import NoiseScheduler
    from stdiff_amd.engine import step_coefficients
    ns = NoiseScheduler(50, beta_schedule="cosine")
    tb = O.make_tables(50, schedule="cosine")
    coef = step_coefficients(ns).to(DEV)
    R, G, t = 19, 33, 20
    x, e, z, gt = rnd(R, G, seed=1), rnd(R, G, seed=2), rnd(R, G, seed=3), rnd(R, G, seed=4)
    mask = (torch.arange(G) % 3 != 0).float()
    prev, _ = O.step(tb, e, t, x, z)
    ref = prev * (1.0 - mask) + mask * gt
    xd = x.to(DEV)
    xp = torch.zeros(R, 64, device=DEV, dtype=torch.bfloat16)
    tdev = torch.tensor([t], dtype=torch.int32, device=DEV)
    ops.ddpm_step(xd, e.to(DEV), gt.to(DEV), mask.to(DEV), coef, tdev, t, z_base=z.to(DEV), xp=xp)
    assert np.array_equal(xd.cpu().numpy(), ref.numpy())
    assert torch.equal(xp[:, :G].cpu(), ref.to(torch.bfloat16)) and float(xp[:, G:].abs().max()) == 0
    ops.advance_timestep(tdev)
    assert int(tdev.item()) == t - 1


def test_philox_noise_statistics(ops):
    """In-kernel Philox + Box-Muller: N(0,1) moments, independence of row offset blocks, determinism."""
    R, G = 4096, 256
    gt, mask = torch.zeros(R, G, device=DEV), torch.zeros(G, device=DEV)
    a, b, c = (torch.empty(R, G, device=DEV) for _ in range(3))
    ops.init_sample(a, gt, mask, seed=1234)
    ops.init_sample(b, gt, mask, seed=1234)
    ops.init_sample(c[:R // 2], gt[:R // 2], mask, seed=1234, row_offset=R // 2)
    assert torch.equal(a, b)
    assert torch.equal(c[:R // 2], a[R // 2:])            # world-size invariance: keyed by GLOBAL row
    v = a.double().cpu()
    assert abs(v.mean()) < 5e-3 and abs(v.std() - 1) < 5e-3
    assert abs((v ** 3).mean()) < 2e-2 and abs((v ** 4).mean() - 3) < 5e-2
    assert abs(float(torch.corrcoef(torch.stack([v[:-1].flatten(), v[1:].flatten()]))[0, 1])) < 5e-3


def test_masked_mse(ops):
    R, G = 70, 40
    pred, noise = rnd(R, G, seed=1), rnd(R, G, seed=2)
    mask = (torch.arange(G) % 5 != 0).float()
    p = pred.clone().requires_grad_(True)
    loss = F.mse_loss(noise * (1 - mask), p * (1 - mask))
    loss.backward()
    l = torch.zeros(1, device=DEV)
    d = torch.empty(R, G, device=DEV)
    ops.masked_mse(pred.to(DEV), noise.to(DEV), mask.to(DEV), l, dpred=d)
    assert abs(float(l) - float(loss)) < 1e-6 and relerr(d, p.grad) < 1e-6


@pytest.mark.parametrize("tdt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M", [1, 100, 512, 2048 + 37])
def test_gemm_ln_mod_epilogue_equals_unfused(tdt, M):
    """STDIFF_EPI_LN_MOD: the adaLN projection whose epilogue applies the LayerNorm + modulate it feeds must give, bit
    for bit, what the plain GEMM into a 16-bit [M, 2H] buffer followed by stdiff_ln_modulate gives (and that pair is
    checked against torch below)."""
    from stdiff_amd import _lib as L
    from stdiff_amd import ops
    H, K = 256, 128
    g = torch.Generator().manual_seed(M)
    s = torch.randn(M, K, generator=g).to(DEV).to(tdt)
    W = (torch.randn(2 * H, K, generator=g) * K ** -0.5).to(DEV).to(tdt)      # rows [shift(H) ; scale(H)]
    b = torch.randn(2 * H, generator=g).to(DEV)
    h = (torch.randn(M, H, generator=g) * 1.7 + 0.3).to(DEV)
    # unfused: mod = s W^T + b (16-bit), xn = LN(h) * (1 + scale) + shift
    mod = ops.gemm(s, W, b, out_dtype=tdt)
    want = ops.ln_modulate(h, mod[:, :H], mod[:, H:], out_dtype=tdt)
    # fused: rows permuted into 256-row tiles [128 shift | 128 scale]
    t = torch.arange(H // 128, device=DEV).repeat_interleave(256) * 128
    j = torch.arange(256, device=DEV).repeat(H // 128)
    perm = t + (j % 128) + (j // 128) * H
    parts = ops.row_stats(h)                              # [M, 1, 2] = (mean, M2): the LN kernel's own reduction
    got = torch.full((M, H), float("nan"), device=DEV, dtype=tdt)
    Wp, bp = W[perm].contiguous(), b[perm].contiguous()
    ops.gemm(s, Wp, bp, out=got, epi=L.EPI_LN_MOD, res=h, ln_parts=parts)
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    ref = torch.nn.functional.layer_norm(h, (H,), eps=1e-6) * (1 + mod[:, H:].float()) + mod[:, :H].float()
    assert relerr(got, ref) < (1e-2 if tdt == torch.bfloat16 else 2e-3)
    assert relerr(parts[:, 0, 0], h.mean(1)) < 1e-5 and relerr(parts[:, 0, 1], h.var(1, unbiased=False) * H) < 1e-5
    with pytest.raises(ValueError):                    # N must be a multiple of 256
        ops.gemm(s, Wp[:384].contiguous(), bp[:384].contiguous(), out=got, epi=L.EPI_LN_MOD, res=h, ln_parts=parts)
    # the statistics handed over by the epilogue that WRITES the residual stream (stats_out of a gated-residual GEMM:
    # one (mean_p, M2_p) pair per row and 256-column tile) give the same LayerNorm to rounding
    A2 = torch.randn(M, K, generator=g).to(DEV).to(tdt)
    W2 = (torch.randn(H, K, generator=g) * K ** -0.5).to(DEV).to(tdt)
    gate = torch.randn(M, H, generator=g).to(DEV).to(tdt)
    h2, h2b = torch.empty(M, H, device=DEV), torch.empty(M, H, device=DEV)
    so = torch.full((M, H // 256, 2), float("nan"), device=DEV)
    ops.gemm(A2, W2, None, out=h2, epi=L.EPI_GATE_RES, gate=gate, res=h, stats_out=so)
    ops.gemm(A2, W2, None, out=h2b, epi=L.EPI_GATE_RES, gate=gate, res=h)
    assert torch.equal(h2, h2b)                           # producing the statistics does not touch the output
    cols = h2.view(M, H // 256, 256)
    assert relerr(so[:, :, 0], cols.mean(2)) < 1e-5 and relerr(so[:, :, 1], cols.var(2, unbiased=False) * 256) < 1e-4
    got2 = torch.empty((M, H), device=DEV, dtype=tdt)
    ops.gemm(s, Wp, bp, out=got2, epi=L.EPI_LN_MOD, res=h2, ln_parts=so)
    want2 = ops.ln_modulate(h2, mod[:, :H], mod[:, H:], out_dtype=tdt)
    assert relerr(got2, want2) < (8e-3 if tdt == torch.bfloat16 else 1e-3)          # one 16-bit ulp of the largest value
