"""Which buffer placements make the fc1 GEMM (65536 x 4096 x 1024, GELU, bf16) slow?  Run as the FIRST GPU process on a
fresh box: buffers carved from one arena at chosen byte offsets."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stdiff_amd import ops, _lib as L
dev = torch.device('cuda:0')
M, N, K = 65536, 4096, 1024
arena = torch.empty(4 << 30, device=dev, dtype=torch.uint8)
base = arena.data_ptr()
print("arena base % 1GiB =", hex(base % (1 << 30)))
W = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
b = torch.randn(N, device=dev)
MB = 1 << 20

def view(off, rows, cols, ld=None):
    ld = ld or cols
    t = arena[off:off + rows * ld * 2].view(torch.bfloat16).view(rows, ld)
    return t[:, :cols]

def run(a_off, c_off, ldc=None, tag=""):
    A = view(a_off, M, K)
    C = view(c_off, M, N, ldc)
    A.normal_(0, 0.5)
    for _ in range(2):
        ops.gemm(A, W, b, out=C, epi=L.EPI_GELU)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.gemm(A, W, b, out=C, epi=L.EPI_GELU)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 5 * 1e3
    print(f"A@{a_off / MB:8.2f}MB C@{c_off / MB:8.2f}MB ldc={ldc or N} {tag}: {us:7.1f} us = {2.0 * M * N * K / us / 1e6:6.0f} TF", flush=True)

# the layout of Workspace in a fresh process: h 256MB, s 128, mod 768, xn 128 @1152, qkv 384, o 128, u 512 @1792
run(1152 * MB, 1792 * MB, tag="workspace layout")
run(0, 128 * MB, tag="A,C adjacent aligned")
run(0, 1024 * MB, tag="C at 1 GiB")
run(0, 128 * MB + 4096, tag="C +4KB")
run(0, 128 * MB + 65536, tag="C +64KB")
run(0, 129 * MB, tag="C +1MB")
run(0, 130 * MB, tag="C +2MB")
run(0, 136 * MB, tag="C +8MB")
run(0, 160 * MB, tag="C +32MB")
run(0, 192 * MB, tag="C +64MB")
run(4096, 128 * MB, tag="A +4KB")
run(0, 128 * MB, ldc=N + 64, tag="ldc padded 128B")
run(0, 128 * MB, ldc=N + 512, tag="ldc padded 1KB")
run(1152 * MB, 1792 * MB, tag="workspace layout again")
