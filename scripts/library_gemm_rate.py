"""Measurement only (not a product path, nothing imports it): the rate of the vendor GEMM library behind torch.mm at
the sampler's shapes, next to stdiff_gemm at the same shapes.  It answers one question for DESIGN.md: how far is the
hand-written kernel from what a tuned library reaches on this GPU at these sizes?"""
import sys
import time

import torch

sys.path.insert(0, ".")
from stdiff_amd import ops  # noqa: E402


def rate(fn, flop, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return ms * 1e3, flop / ms / 1e9


def main():
    dev = torch.device("cuda:0")
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
    shapes = [(1024, 3072), (1024, 1024), (1024, 4096), (4096, 1024), (1024, 6144)]
    for dt in (torch.bfloat16, torch.float16):
        for K, N in shapes:
            a = torch.randn(M, K, device=dev).to(dt)
            w = (torch.randn(N, K, device=dev) * 0.03).to(dt)
            bias = torch.randn(N, device=dev)
            flop = 2.0 * M * N * K
            us_l, tf_l = rate(lambda: torch.nn.functional.linear(a, w), flop)
            line = f"{str(dt)[6:]:9s} M={M} K={K} N={N}: library {us_l:8.1f} us {tf_l:7.1f} TFLOP/s"
            if dt == torch.bfloat16:
                c = torch.empty(M, N, device=dev, dtype=dt)
                us_o, tf_o = rate(lambda: ops.gemm(a, w, bias, c), flop)
                line += f" | stdiff_gemm(+bias) {us_o:8.1f} us {tf_o:7.1f} TFLOP/s"
            print(line, flush=True)
            del a, w
    time.sleep(0.1)


if __name__ == "__main__":
    main()
