"""gemm192t_kernel (two tiles in flight, STDIFF_GEMM_TILE=5) against gemm256q_kernel (STDIFF_GEMM_TILE=4): the same bits,
then HIP-event times at the sampler's shapes.   python scripts/gemm192_check.py [rows] [dtype]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stdiff_amd import _lib as L, ops
dev = "cuda:0"
R = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
tdt = torch.float16 if (len(sys.argv) > 2 and sys.argv[2] == "fp16") else torch.bfloat16
L.lib(tdt == torch.float16)


def run(tile, A, W, b, epi):
    os.environ["STDIFF_GEMM_TILE"] = str(tile)
    L.reload_env()
    out = torch.full((A.shape[0], W.shape[0]), float("nan"), device=dev, dtype=tdt)
    ops.gemm(A, W, b, out=out, epi=epi)
    torch.cuda.synchronize()
    return out


g = torch.Generator(device=dev).manual_seed(0)
bad = 0
for (M, N, K) in [(192, 256, 832), (1000, 520, 832), (4096, 2048, 1024), (777, 1032, 4096), (40000, 3072, 1024)]:
    A = torch.randn(M, K, device=dev, generator=g).to(tdt)
    W = (torch.randn(N, K, device=dev, generator=g) * K ** -0.5).to(tdt)
    b = torch.randn(N, device=dev, generator=g)
    for epi, name in ((L.EPI_BIAS, "bias"), (L.EPI_GELU, "gelu")):
        for bias in (b, None):
            ref = run(4, A, W, bias, epi)
            got = run(5, A, W, bias, epi)
            same = torch.equal(ref.view(torch.int16), got.view(torch.int16))
            nbad = int((ref.view(torch.int16) != got.view(torch.int16)).sum())
            print(f"M={M} N={N} K={K} {name} bias={'y' if bias is not None else 'n'}: {'equal' if same else f'DIFFERENT in {nbad} elements'}"
                  f" (finite {bool(torch.isfinite(got.float()).all())})", flush=True)
            if not same:
                bad += 1
                d = (ref.float() - got.float()).abs()
                idx = torch.nonzero(ref.view(torch.int16) != got.view(torch.int16))[:6]
                print("   first differing (row, col):", idx.tolist(), "max |d|", float(d.nan_to_num(1e9).max()))
if bad:
    print(f"RESULT: {bad} mismatching cases")
    sys.exit(1)
print("RESULT: bit-equal everywhere")
# ---- timing at the sampler's shapes
H = 1024
t0 = time.time()
X = torch.randn(R, H, device=dev, generator=g).to(tdt)
while time.time() - t0 < 2.0:
    (X.float()[:8192] @ X.float()[:8192].T).sum().item()
for name, N, K, epi in (("gates", 2 * H, H, L.EPI_BIAS), ("qkv", 3 * H, H, L.EPI_BIAS), ("fc1", 4 * H, H, L.EPI_GELU)):
    A = torch.randn(R, K, device=dev, generator=g).to(tdt) * 0.5
    W = (torch.randn(N, K, device=dev, generator=g) * K ** -0.5).to(tdt)
    b = torch.randn(N, device=dev, generator=g)
    out = torch.empty(R, N, device=dev, dtype=tdt)
    res = {}
    for tile in (4, 5, 4, 5):
        os.environ["STDIFF_GEMM_TILE"] = str(tile)
        L.reload_env()
        for _ in range(3):
            ops.gemm(A, W, b, out=out, epi=epi)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.gemm(A, W, b, out=out, epi=epi)
        e1.record()
        torch.cuda.synchronize()
        res.setdefault(tile, []).append(e0.elapsed_time(e1) / 10 * 1e3)
    fl = 2.0 * R * N * K
    print(f"{name:6s} R={R}: 256q {min(res[4]):8.1f} us ({fl / min(res[4]) / 1e6:6.0f} TF)   192t {min(res[5]):8.1f} us "
          f"({fl / min(res[5]) / 1e6:6.0f} TF)   ratio {min(res[4]) / min(res[5]):.3f}", flush=True)
