"""Diagnostic for DESIGN.md 6a', second stage: is it the NUMBER of hipGraphLaunch calls (or the depth of the launch
queue) of a parent that makes new GPU processes die in hsa_init?

    python scripts/diag/launch_leak.py <mode> [rounds] [steps_per_round]

modes: reuse (one 2-lane sampler, its two graphs replayed over and over), reuse_sync (the same, host sync every 32
steps), fresh (a new sampler -- new graphs, new streams -- every round), reuse_1lane, eager (no graph, one stream).
After every round a child runs hsa_init (scripts/diag/hsa_probe); the first failing round is reported.
"""
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from stdiff_amd import engine as E  # noqa: E402
from stdiff_amd import ops  # noqa: E402
from stdiff_amd import synth as O  # noqa: E402
from stdiff_amd.model.stDiff_model import DiT_stDiff  # noqa: E402
from stdiff_amd.model.stDiff_scheduler import NoiseScheduler  # noqa: E402

PROBE = os.path.join(ROOT, "scripts", "diag", "hsa_probe")
N_LAUNCH = [0]
_launch = ops.Graph.launch


def launch(self):
    N_LAUNCH[0] += 1
    _launch(self)


ops.Graph.launch = launch


def probe():
    r = subprocess.run([PROBE], capture_output=True, text=True)
    return r.returncode == 0


def main():
    mode = sys.argv[1]
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    dev = "cuda:0"
    N, G, H, depth, heads, B, T = 10000, 512, 1024, 6, 16, 512, 1000
    torch.manual_seed(0)
    m = DiT_stDiff(G, H, depth, 'dit', heads, 6, compute_dtype="bf16").to(dev).eval()
    mask = torch.from_numpy(O.kfold_mask(G, 0)).to(dev)
    gt = torch.from_numpy(O.synth_expression(N, G, 1)).to(dev) * mask * 2 - 1
    sched = NoiseScheduler(T, beta_schedule="cosine")
    lens = [min(B, N - i) for i in range(0, N, B)]
    lanes = 1 if mode in ("reuse_1lane", "fresh_1lane") else 2
    graph = mode not in ("eager", "fresh_eager")
    if mode == "fresh_cached_streams":                   # every sampler gets the SAME two lane streams
        pair = E.concurrent_streams(dev, 2)
        E.concurrent_streams = lambda device, n, **kw: pair[:n]
    if mode == "fresh_1step":
        steps = 1

    def make():
        s = E.LanedSampler(m.engine(), sched, lens, gt, gt, mask, seed=1, lanes=lanes, use_graph=graph)
        s.init()
        return s
    assert probe(), "probe fails before anything ran"
    smp = make()
    ts = list(range(T))[::-1]
    with torch.no_grad():
        for r in range(rounds):
            if mode.startswith("fresh") and r:
                smp = make()
            if mode == "fresh_norun":
                torch.cuda.synchronize()
                ok = probe()
                print(f"{mode}: round {r + 1}: probe {'ok' if ok else 'FAILED'}", flush=True)
                if not ok:
                    print(f"RESULT {mode}: first failure after {r + 1} samplers (never run)", flush=True)
                    return
                continue
            if mode == "reuse_sync":
                smp.run(ts[:32])
                done = 32
                while done < steps:
                    torch.cuda.synchronize()
                    k = min(32, steps - done)
                    smp.run_more(k)
                    done += k
            else:
                smp.run(ts[:steps])
            torch.cuda.synchronize()
            ok = probe()
            print(f"{mode}: round {r + 1}: graph launches so far {N_LAUNCH[0]}, probe {'ok' if ok else 'FAILED'}", flush=True)
            if not ok:
                print(f"RESULT {mode}: first failure after {N_LAUNCH[0]} graph launches ({r + 1} rounds of {steps} steps)", flush=True)
                return
    print(f"RESULT {mode}: no failure after {N_LAUNCH[0]} graph launches ({rounds} rounds of {steps} steps)", flush=True)


if __name__ == "__main__":
    main()
