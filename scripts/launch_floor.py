"""Measurement only: device time per node of a hipGraph chain of tiny dependent kernels (the floor any launch of the
training chain pays), and of a single 128 x 128 x K GEMM tile sequence for comparison."""
import sys
import torch
sys.path.insert(0, ".")
from stdiff_amd import ops  # noqa: E402
from scripts.small_gemm_time import timed  # noqa: E402

dev = torch.device("cuda:0")
x = torch.zeros(8, 64, device=dev)
y = torch.zeros(8, 64, device=dev, dtype=torch.bfloat16)
print(f"tiny kernel (pack_rows 8x64): {timed(lambda: ops.pack_rows(x, 64, torch.bfloat16, out=y)):.2f} us per graph node")
print(f"spin_probe(1 block, 1 iteration): {timed(lambda: ops.spin_probe(1, 1)):.2f} us per graph node")
bf = torch.bfloat16
for M, N, K in [(128, 128, 64), (128, 128, 1024), (512, 512, 1024), (512, 3072, 64), (512, 3072, 1024)]:
    a = torch.randn(M, K, device=dev).to(bf)
    w = torch.randn(N, K, device=dev).to(bf)
    c = torch.empty(M, N, device=dev, dtype=bf)
    print(f"gemm {M}x{N}x{K}: {timed(lambda: ops.gemm(a, w, None, c)):.2f} us per graph node")
for M, N, K in [(128, 128, 4096), (128, 128, 16384)]:
    a = torch.randn(M, K, device=dev).to(bf)
    w = torch.randn(N, K, device=dev).to(bf)
    c = torch.empty(M, N, device=dev, dtype=bf)
    print(f"gemm {M}x{N}x{K}: {timed(lambda: ops.gemm(a, w, None, c)):.2f} us per graph node")
