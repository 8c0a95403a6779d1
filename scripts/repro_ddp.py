"""Repro: a diffusion_impute call in the parent, then the 2-rank DDP golden workers (tests/test_gpu_train.py)."""
import os, sys, socket, tempfile, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import cpu_ref as O

def main():
    env0 = dict(os.environ)
    what = sys.argv[1] if len(sys.argv) > 1 else "impute"
    DEV = "cuda:0"
    if what == "impute":
        from stdiff_amd.driver import diffusion_impute
        N, G, H, depth, heads, B, T = int(sys.argv[3]), 500, 1024, 6, 16, 512, int(sys.argv[4])
        nfold = int(sys.argv[5]) if len(sys.argv) > 5 else 5
        seq = O.synth_expression(1024, G, 21); st = O.synth_expression(N, G, 22)
        with tempfile.TemporaryDirectory() as ck:
            for k in range(5):
                torch.save(O.make_params(G, H, depth, 60 + k), f"{ck}/stDiff{k + 1}.pt")
            for dt in sys.argv[2].split(","):
                for k in range(nfold):
                    r = diffusion_impute(seq, st, depth=depth, num_epoch=1, diffusion_step=T, batch_size=B, hidden_size=H, head=heads,
                                         device=torch.device(DEV), save_dir=ck, sample_seed=1, compute_dtype=dt, folds=[k])
                    print("impute", dt, k, float(np.abs(r).sum()), torch.cuda.memory_allocated() >> 20, torch.cuda.memory_reserved() >> 20, 'free MB', torch.cuda.mem_get_info()[0] >> 20, 'fds', len(os.listdir('/proc/self/fd')), 'threads', len(os.listdir('/proc/self/task')), flush=True)
    import subprocess
    print('ENV DIFF', {k: (env0.get(k), os.environ.get(k)) for k in set(env0) | set(os.environ) if env0.get(k) != os.environ.get(k)}, flush=True)
    r = subprocess.run([sys.executable, '-c', "import torch; print('plain child, original env ok', float(torch.ones(4).cuda().sum()))"], capture_output=True, text=True, env=env0)
    print('PLAIN CHILD ENV0 rc', r.returncode, r.stdout.strip()[-40:], flush=True)
    r = subprocess.run(['rocminfo'], capture_output=True, text=True)
    print('ROCMINFO rc', r.returncode, r.stdout.strip()[-200:].replace(chr(10), ' | '), r.stderr.strip()[-200:], flush=True)
    r = subprocess.run([sys.executable, '-c', "import faulthandler; faulthandler.enable(); import torch; print('plain child ok', float(torch.ones(4).cuda().sum()))"], capture_output=True, text=True)
    print('PLAIN CHILD rc', r.returncode, r.stdout.strip()[-40:], flush=True)
    r = subprocess.run([sys.executable, '-c', "import faulthandler; faulthandler.enable(); import torch, os, sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests'); import test_gpu_train as TT; m, p = TT.build(40, 128, 2, 4, 16, 'fp32'); print('build child ok')"], capture_output=True, text=True, cwd=ROOT)
    print('BUILD CHILD rc', r.returncode, r.stdout.strip()[-40:], r.stderr.strip()[-300:], flush=True)
    import test_gpu_train as TT
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    t0 = time.time()
    procs = [ctx.Process(target=TT._ddp_worker, args=(r, 2, port, os.path.join(ROOT, "tests", "golden"), q, "gloo")) for r in range(2)]
    for p in procs: p.start()
    print("workers started", flush=True)
    try:
        res = [q.get(timeout=60) for _ in procs]
        print("workers done", time.time() - t0, [(r[0], r[1]) for r in res], flush=True)
    except Exception as e:
        print("TIMEOUT waiting for workers", repr(e), [p.is_alive() for p in procs], flush=True)
        for p in procs: p.kill()
    for p in procs: p.join(10)

if __name__ == "__main__":
    main()
