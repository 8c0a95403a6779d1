"""Measurement only: device time per launch of the training step's batch-sized GEMMs (M = cells), taken from a hipGraph
of REP back-to-back launches so that host launch cost is out of the picture.
usage: python scripts/small_gemm_time.py [M] ; knobs through the environment (STDIFF_GEMM_DEPTH=1|3)"""
import sys

import torch

sys.path.insert(0, ".")
from stdiff_amd import ops  # noqa: E402

REP = 40


def timed(fn):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        g = ops.Graph()
        g.begin()
        for _ in range(REP):
            fn()
        g.end()
        for _ in range(3):
            g.launch()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            g.launch()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * REP)


def main():
    dev = torch.device("cuda:0")
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    ws = torch.empty(64 << 20, device=dev, dtype=torch.uint8)
    bf = torch.bfloat16
    for K, N in [(1024, 3072), (1024, 1024), (1024, 4096), (4096, 1024), (1024, 6144)]:
        a = torch.randn(M, K, device=dev).to(bf)
        w = (torch.randn(N, K, device=dev) * 0.03).to(bf)
        dy = torch.randn(M, N, device=dev).to(bf)
        bias = torch.randn(N, device=dev)
        c = torch.empty(M, N, device=dev, dtype=bf)
        dx = torch.empty(M, K, device=dev, dtype=bf)
        dw = torch.empty(N, K, device=dev)
        fl = 2.0 * M * N * K
        t_f = timed(lambda: ops.gemm(a, w, bias, c, workspace=ws))
        t_x = timed(lambda: ops.gemm(dy, w, None, dx, workspace=ws, w_kmajor=True))          # dX = dY W
        t_w = timed(lambda: ops.gemm(dy, a, None, dw, workspace=ws, a_kmajor=True, w_kmajor=True))   # dW = dY^T X
        print(f"M={M} K={K} N={N}: fwd {t_f:6.1f} us ({fl / t_f / 1e6:6.0f} TF/s)  dX {t_x:6.1f} us ({fl / t_x / 1e6:6.0f})"
              f"  dW {t_w:6.1f} us ({fl / t_w / 1e6:6.0f})", flush=True)


if __name__ == "__main__":
    main()
