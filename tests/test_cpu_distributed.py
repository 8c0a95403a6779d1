"""CPU, world_size 2 over gloo: the N>1 paths that do not need a GPU -- chunk-aligned shard plan + gather of the
sampling path, and the bucketed flat-gradient all-reduce of data-parallel training, checked against
golden G7 (reference gradients of two batches averaged, one clip + AdamW step)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import cpu_ref as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    try:
        from stdiff_amd.parallel import gather_rows, shard_chunks
        from stdiff_amd.train_engine import flat_layout, bucket_ranges
        # ---- sampling: shard plan + gather reproduces the row order
        lens = [40, 40, 40, 40, 17]
        lo, hi, mine = shard_chunks(lens, rank, world)
        assert sum(mine) == hi - lo
        local = np.arange(lo, hi, dtype=np.float32)[:, None] * np.ones((1, 3), dtype=np.float32)
        full = gather_rows(local, sum(lens))
        assert np.array_equal(full[:, 0], np.arange(sum(lens), dtype=np.float32))

        # ---- training: each rank = the reference step on ITS batch; flat grads all-reduced bucket by bucket
        g6 = np.load(os.path.join(GOLDEN, "g6_train.npz"))
        g7 = np.load(os.path.join(GOLDEN, "g7_ddp.npz"))
        C, G, H, depth, heads, seed, T = [int(v) for v in g6["cfg"]]
        p = O.make_params(G, H, depth, seed)
        tb = O.make_tables(T, schedule="cosine")
        x = torch.from_numpy(g6["x"])[rank * C:(rank + 1) * C]
        xc = torch.from_numpy(g6["x_cond"])[rank * C:(rank + 1) * C]
        _, grads, _ = O.loss_and_grads(p, tb, x, xc, torch.from_numpy(g6["noise"][rank]),
                                       torch.from_numpy(g6["timesteps"][rank]), g6["mask"], heads, depth)
        names = O.trainable_names(p)
        offs, total = flat_layout([(n, p[n].numel()) for n in names])
        flat = torch.zeros(total)
        for n in names:
            flat[offs[n]:offs[n] + p[n].numel()] = grads[n].reshape(-1)
        ranges = bucket_ranges(offs, total, depth)
        cover = torch.zeros(total, dtype=torch.int32)
        for lo_, hi_ in ranges:
            cover[lo_:hi_] += 1
        assert bool((cover == 1).all())                      # buckets tile the flat buffer exactly once
        hs = [dist.all_reduce(flat[lo_:hi_], async_op=True) for lo_, hi_ in ranges]
        for h in hs:
            h.wait()
        avg = {n: (flat[offs[n]:offs[n] + p[n].numel()] / world).view_as(p[n]).clone() for n in names}
        gn = O.clip_grads_(avg, 1.0)
        O.adamw_step_(p, avg, {}, float(g6["lr"][0]))
        assert abs(float(gn) - g7["gnorm"][0]) < 1e-5 * g7["gnorm"][0]
        for k in g7.files:
            if k.startswith("post/"):
                tol = 5e-5 if k.endswith("qkv.bias") else 1e-5
                d = float((p[k[5:]] - torch.from_numpy(g7[k])).abs().max()) / float(np.abs(g7[k]).max())
                assert d < tol, (k, d)
        q.put((rank, "ok"))
    except Exception as e:      # surface the failure in the parent
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(r[1] == "ok" for r in res), res
