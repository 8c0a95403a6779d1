"""In-kernel time stamps of the persistent 256x256 GEMM (wait / K loop / epilogue per tile, per-quarter epilogue detail,
shader clock over the K loop).  Needs a debug build of gemm.hip:
    hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -DSTDIFF_GEMM_STAMPS -shared \
          -o scratch/libgemm_dbg.so stdiff_amd/csrc/gemm.hip
run from the repo root on the GPU box: TILE=4 python tools/gemm_stamps.py   (TILE=3: the 4-phase kernel)."""
import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, '.')
from stdiff_amd import _lib as L
dbg = ctypes.CDLL(os.path.abspath("scratch/libgemm_dbg.so"))
dbg.stdiff_gemm.restype = ctypes.c_int
dbg.stdiff_gemm.argtypes = [ctypes.POINTER(L.GemmArgs), ctypes.c_void_p]
M = 20000
for N, K, epi in ((4096, 1024, 2), (4096, 256, 0), (1024, 4096, 0), (6144, 1024, 0)):
    A = torch.randn(M, K, device='cuda').to(torch.bfloat16)
    W = (torch.randn(N, K, device='cuda') * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    a = L.GemmArgs(M=M, N=N, K=K, A=A.data_ptr(), lda=K, a_dtype=1, W=W.data_ptr(), ldw=K, bias=b.data_ptr(),
                   C=out.data_ptr(), ldc=N, c_dtype=1, epi=epi)
    os.environ["STDIFF_GEMM_TILE"] = os.environ.get("TILE", "3")
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(400):
        assert dbg.stdiff_gemm(ctypes.byref(a), st) == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); dbg.stdiff_gemm(ctypes.byref(a), st); e1.record(); torch.cuda.synchronize()
    buf = np.zeros(256 * 8 * 8, dtype=np.uint64)
    assert dbg.stdiff_debug_gemm_stamps(buf.ctypes.data_as(ctypes.POINTER(ctypes.c_ulonglong))) == 0
    s8 = buf.reshape(256, 8, 8).astype(np.int64); s = s8[:, :, :4]; c = s8[:, :, 4:]
    ntile = (M + 255) // 256 * ((N + 255) // 256)
    iters = [min(8, (ntile - b_ + 255) // 256) for b_ in range(256)]
    t00 = min(s[b_, 0, 0] for b_ in range(256))
    print(f"N={N} K={K} epi={epi}: event {e0.elapsed_time(e1)*1e3:.1f} us, tiles {ntile}; ticks are 10 ns")
    wait, loop, ep = [], [], []
    for b_ in range(256):
        for i in range(iters[b_]):
            wait.append(s[b_, i, 1] - s[b_, i, 0]); loop.append(s[b_, i, 2] - s[b_, i, 1]); ep.append(s[b_, i, 3] - s[b_, i, 2])
    clk = np.median([ (c[b_, i, 2] - c[b_, i, 1]) / max(1, (s[b_, i, 2] - s[b_, i, 1])) * 100e6 for b_ in range(256) for i in range(iters[b_]) ])
    print(f"   in-kernel clock over the K loop (median): {clk/1e9:.3f} GHz; K-loop cycles per slab {np.median([(c[b_, i, 2] - c[b_, i, 1]) for b_ in range(256) for i in range(iters[b_])]) / (K // 64):.0f}")
    wait, loop, ep = np.array(wait) / 100., np.array(loop) / 100., np.array(ep) / 100.
    print(f"   wait-operands us: mean {wait.mean():.2f} p50 {np.median(wait):.2f} max {wait.max():.2f}")
    print(f"   k-loop us:        mean {loop.mean():.2f} p50 {np.median(loop):.2f} max {loop.max():.2f}")
    print(f"   epilogue us:      mean {ep.mean():.2f} p50 {np.median(ep):.2f} max {ep.max():.2f}")
    end = max(s[b_, iters[b_] - 1, 3] for b_ in range(256))
    starts = np.array([s[b_, 0, 0] for b_ in range(256)]) - t00
    ends = np.array([s[b_, iters[b_] - 1, 3] for b_ in range(256)]) - t00
    print(f"   first-start spread {starts.max()/100.:.2f} us; kernel span {(end-t00)/100.:.2f} us; block end p50 {np.median(ends)/100.:.2f} min {ends.min()/100.:.2f}")
    for b_ in ():
        print("   block", b_, [[int(x - t00) for x in s[b_, i]] for i in range(iters[b_])])
    eb = np.zeros(256 * 8 * 8, dtype=np.uint64)
    assert dbg.stdiff_debug_gemm_estamps(eb.ctypes.data_as(ctypes.POINTER(ctypes.c_ulonglong))) == 0
    e = eb.reshape(256, 8, 8).astype(np.int64)
    rows = np.array([np.concatenate(([s[b_, i, 2]], e[b_, i], [s[b_, i, 3]])) for b_ in range(256) for i in range(iters[b_])])
    d = np.diff(rows, axis=1) / 100.
    print("   epilogue detail us (median): kloop_end->img0 " + " | ".join(f"{np.median(d[:, k]):.2f}" for k in range(d.shape[1])),
          "  [alternating: put+barrier, image->stores issued]")
