"""In-kernel time stamps of the persistent 256x256 GEMM (wait / K loop / epilogue per tile; shader clock over the K
loop).  Needs the debug build: hipcc ... -DSTDIFF_GEMM_STAMPS -shared -o stdiff_amd/csrc/build/dbg/libgemm_dbg.so gemm.hip"""
import ctypes, os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stdiff_amd import _lib as L
dbg = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "stdiff_amd/csrc/build/dbg/libgemm_dbg.so"))
dbg.stdiff_gemm.restype = ctypes.c_int
dbg.stdiff_gemm.argtypes = [ctypes.POINTER(L.GemmArgs), ctypes.c_void_p]
M = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
NB = int(os.environ.get('STDIFF_GEMM_CUS', '256'))
H = 1024
only = sys.argv[2].split(",") if len(sys.argv) > 2 else None
for name, N, K, epi, cdt in (("ada", 6144, 1024, 0, 1), ("fc1", 4096, 1024, 2, 1), ("proj", 1024, 1024, 3, 0), ("fc2", 1024, 4096, 3, 0),
                             ("ln", 2048, 1024, 7, 1)):
    if only and name not in only:
        continue
    A = torch.randn(M, K, device='cuda').to(torch.bfloat16)
    W = (torch.randn(N, K, device='cuda') * K ** -0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    out = torch.empty(M, N // 2 if epi == 7 else N, device='cuda', dtype=torch.bfloat16 if cdt else torch.float32)
    gate = torch.randn(M, 6 * H, device='cuda').to(torch.bfloat16)
    a = L.GemmArgs(M=M, N=N, K=K, A=A.data_ptr(), lda=K, a_dtype=1, W=W.data_ptr(), ldw=K, bias=b.data_ptr(),
                   C=out.data_ptr(), ldc=out.shape[1], c_dtype=cdt, epi=epi)
    if epi == 7:            # LN_MOD: residual stream + 4 partial moments per row
        hres = torch.randn(M, N // 2, device='cuda')
        parts = torch.stack([hres.view(M, 4, -1).mean(-1), hres.view(M, 4, -1).var(-1, unbiased=False) * (N // 8)], -1).contiguous()
        a.res, a.ldres, a.ln_parts, a.ln_nparts = hres.data_ptr(), N // 2, parts.data_ptr(), 4
    if epi == 3:
        a.gate, a.ldgate, a.res, a.ldres = gate.data_ptr(), 6 * H, out.data_ptr(), N
    st = torch.cuda.current_stream().cuda_stream
    t0 = time.time()
    while time.time() - t0 < 1.5:
        for _ in range(20):
            assert dbg.stdiff_gemm(ctypes.byref(a), st) == 0
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); dbg.stdiff_gemm(ctypes.byref(a), st); e1.record(); torch.cuda.synchronize()
    buf = np.zeros(256 * 8 * 8, dtype=np.uint64)
    assert dbg.stdiff_debug_gemm_stamps(buf.ctypes.data_as(ctypes.POINTER(ctypes.c_ulonglong))) == 0
    s8 = buf.reshape(256, 8, 8).astype(np.int64); s = s8[:, :, :4]; c = s8[:, :, 4:]
    ntile = (M + 255) // 256 * ((N + 255) // 256)
    iters = [min(8, (ntile - b_ + NB - 1) // NB) for b_ in range(NB)]
    us = e0.elapsed_time(e1) * 1e3
    print(f"{name} M={M} N={N} K={K} epi={epi}: event {us:.1f} us = {2.0 * M * N * K / us / 1e6:.0f} TF, tiles {ntile} ({ntile / NB:.1f} per CU); first 8 tiles of each CU:")
    wait, loop, ep, gap = [], [], [], []
    for b_ in range(NB):
        for i in range(iters[b_]):
            wait.append(s[b_, i, 1] - s[b_, i, 0]); loop.append(s[b_, i, 2] - s[b_, i, 1]); ep.append(s[b_, i, 3] - s[b_, i, 2])
            if i: gap.append(s[b_, i, 0] - s[b_, i - 1, 3])
    cyc = np.array([(c[b_, i, 2] - c[b_, i, 1]) for b_ in range(NB) for i in range(iters[b_])])
    clk = np.median(cyc / np.maximum(1, np.array(loop))) * 100e6
    wait, loop, ep = np.array(wait) / 100., np.array(loop) / 100., np.array(ep) / 100.
    print(f"   clock over the K loop (median) {clk / 1e9:.3f} GHz; K-loop cycles per slab {np.median(cyc) / (K // 64):.0f}")
    print(f"   tile-start wait us: mean {wait.mean():.2f} p50 {np.median(wait):.2f} max {wait.max():.2f}")
    print(f"   k-loop us:          mean {loop.mean():.2f} p50 {np.median(loop):.2f} max {loop.max():.2f}")
    print(f"   epilogue us:        mean {ep.mean():.2f} p50 {np.median(ep):.2f} max {ep.max():.2f}")
    print(f"   per tile total us:  mean {(wait + loop + ep).mean():.2f}  (event / tiles per CU = {us / (ntile / NB):.2f})")
    if epi == 7:            # sub-stamps of the LN_MOD epilogue sit in the shader-clock slots (realtime there)
        sub = [[], [], [], []]
        for b_ in range(NB):
            for i in range(iters[b_]):
                t2, t3 = s[b_, i, 2], s[b_, i, 3]
                q4, q5, q6 = c[b_, i, 0], c[b_, i, 1], c[b_, i, 2]
                sub[0].append(q4 - t2); sub[1].append(q5 - q4); sub[2].append(q6 - q5); sub[3].append(t3 - q6)
        sub = [np.array(v) / 100. for v in sub]
        print(f"   LN_MOD epilogue us: loads + statistics {sub[0].mean():.2f}, first half (convert, stage, store) {sub[1].mean():.2f}, "
              f"second half to staged {sub[2].mean():.2f}, its stores {sub[3].mean():.2f}")
