"""Per-phase shader-clock stamps of gemm192t_kernel (debug build, see the STDIFF_T192_STAMPS block in csrc/gemm.hip).
Build:  cd stdiff_amd/csrc && mkdir -p build/dbg && for f in *.hip; do hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 \
        -fno-gpu-rdc -DSTDIFF_T192_STAMPS -c $f -o build/dbg/${f%.hip}.o; done && hipcc --offload-arch=gfx950 -shared -fPIC \
        -Wl,-Bsymbolic -o build/dbg/libstdiff_dbg.so build/dbg/*.o
Run:    STDIFF_LIB=stdiff_amd/csrc/build/dbg/libstdiff_dbg.so python scripts/gemm192_stamps.py [rows] [N] [K]"""
import ctypes as C, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("STDIFF_LIB", os.path.join(ROOT, "stdiff_amd/csrc/build/dbg/libstdiff_dbg.so"))
os.environ["STDIFF_GEMM_TILE"] = "5"
from stdiff_amd import _lib as L, ops
R = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
K = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
dev = "cuda:0"
A = (torch.randn(R, K, device=dev) * 0.5).to(torch.bfloat16)
W = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
b = torch.randn(N, device=dev)
out = torch.empty(R, N, device=dev, dtype=torch.bfloat16)
import time
t0 = time.time()
while time.time() - t0 < 2.0:
    ops.gemm(A, W, b, out=out)
    torch.cuda.synchronize()
ops.gemm(A, W, b, out=out)
torch.cuda.synchronize()
buf = (C.c_ulonglong * (2 * 64 * 10))()
lib = L.lib()
lib.stdiff_debug_t192_stamps.restype = C.c_int
assert lib.stdiff_debug_t192_stamps(buf) == 0
s = np.array(buf, dtype=np.int64).reshape(2, 64, 10)
names = ["P1 issue", "P1 waits", "barrier", "MFMA P1", "barrier", "P2 issue", "waits+bar", "MFMA P2", "barrier"]
for w in range(2):
    d = np.diff(s[w], axis=1)                      # [64, 9]
    slab = s[w, 1:, 0] - s[w, :-1, 0]
    print(f"wave {4 * w} (group {w}): cycles per slab (median over slabs 2..60): {np.median(slab[2:60]):.0f}")
    for tile in range(4):
        sl = slice(16 * tile + (2 if tile == 0 else 0), 16 * tile + 16)
        print(f"  tile {tile}: per-slab {np.median(slab[sl][:-1]) if tile < 3 else np.median(slab[sl][:-2]):.0f}  " +
              "  ".join(f"{n} {np.median(d[sl, i]):.0f}" for i, n in enumerate(names)))
    print("  slab-to-slab:", " ".join(str(int(v)) for v in slab[:34]))
