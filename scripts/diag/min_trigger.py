"""Diagnostic for DESIGN.md 6a', third stage: the smallest thing a parent does that makes new GPU processes die in
hsa_init.  One mode per fresh process, a child probe (scripts/diag/hsa_probe) after every iteration.

    python scripts/diag/min_trigger.py <mode> [iterations]

modes
  graphs_one_stream   capture / instantiate / launch / destroy a one-kernel hipGraph, always on the SAME stream
  streams_only        a tiny kernel on a NEW torch pool stream per iteration (no graph)
  graphs_new_streams  the one-kernel graph on a new pool stream per iteration
  graphs_kept         as graphs_new_streams, the exec handles are kept alive (no hipGraphExecDestroy)
  big_graphs_one_stream   a 600-kernel graph, same stream
  two_streams_concurrent  two one-kernel graphs on two new streams, launched interleaved 50 times (the lanes' pattern)
"""
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from stdiff_amd import ops  # noqa: E402

PROBE = os.path.join(ROOT, "scripts", "diag", "hsa_probe")


def probe():
    return subprocess.run([PROBE], capture_output=True, text=True).returncode == 0


def graph_on(stream, t_dev, kernels=1):
    with torch.cuda.stream(stream):
        g = ops.Graph()
        g.begin()
        try:
            for _ in range(kernels):
                ops.advance_timestep(t_dev)
        finally:
            g.end()
    return g


def main():
    mode = sys.argv[1]
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    dev = "cuda:0"
    t_dev = torch.zeros(1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    assert probe(), "probe fails before anything ran"
    fixed = torch.cuda.Stream(device=dev)
    keep = []
    for i in range(iters):
        if mode == "graphs_one_stream" or mode == "big_graphs_one_stream":
            g = graph_on(fixed, t_dev, 600 if mode.startswith("big") else 1)
            with torch.cuda.stream(fixed):
                g.launch()
            torch.cuda.synchronize()
            del g
        elif mode == "streams_only":
            s = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(s):
                ops.advance_timestep(t_dev)
            torch.cuda.synchronize()
        elif mode in ("graphs_new_streams", "graphs_kept"):
            s = torch.cuda.Stream(device=dev)
            g = graph_on(s, t_dev)
            with torch.cuda.stream(s):
                g.launch()
            torch.cuda.synchronize()
            if mode == "graphs_kept":
                keep.append((s, g))
            del g
        elif mode == "two_streams_concurrent":
            s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
            g1, g2 = graph_on(s1, t_dev, 8), graph_on(s2, t_dev, 8)
            for _ in range(50):
                with torch.cuda.stream(s1):
                    g1.launch()
                with torch.cuda.stream(s2):
                    g2.launch()
            torch.cuda.synchronize()
            del g1, g2
        else:
            raise SystemExit(f"unknown mode {mode}")
        ok = probe()
        if not ok:
            print(f"RESULT {mode}: first failure after iteration {i + 1}", flush=True)
            return
    print(f"RESULT {mode}: no failure in {iters} iterations", flush=True)


if __name__ == "__main__":
    main()
