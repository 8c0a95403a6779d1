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
        full = gather_rows(local, sum(lens))                 # numpy in, numpy out; unequal shards (120 / 57 rows)
        assert isinstance(full, np.ndarray) and np.array_equal(full[:, 0], np.arange(sum(lens), dtype=np.float32))
        tfull = gather_rows(torch.from_numpy(local), sum(lens))          # tensor in, tensor out: ONE padded all_gather
        assert torch.is_tensor(tfull) and torch.equal(tfull, torch.from_numpy(full))
        # the host side of sample_stDiff_sharded: chunk lengths without touching the data, and ONLY this rank's rows
        from torch.utils.data import DataLoader, TensorDataset
        from stdiff_amd.parallel import combine_chunk_sums, loader_chunk_lens, loader_condition_rows
        from stdiff_amd.process.data import get_data_loader
        data = np.arange(sum(lens) * 3, dtype=np.float32).reshape(-1, 3)
        dl = get_data_loader(data, data + 0.5, batch_size=40, is_shuffle=False)
        assert loader_chunk_lens(dl) == lens
        rows = loader_condition_rows(dl, lo, hi, "cpu")
        assert rows.shape == (hi - lo, 3) and torch.equal(rows, torch.from_numpy(data[lo:hi] + 0.5))

        class Odd:                                           # a loader that only iterates (no dataset / sampler)
            def __iter__(self):
                for i in range(0, len(data), 40):
                    yield torch.from_numpy(data[i:i + 40]), torch.from_numpy(data[i:i + 40] + 0.5)
        assert loader_chunk_lens(Odd()) == lens
        assert torch.equal(loader_condition_rows(Odd(), lo, hi, "cpu"), rows)
        dl_drop = DataLoader(TensorDataset(torch.from_numpy(data), torch.from_numpy(data)), batch_size=40, drop_last=True)
        assert loader_chunk_lens(dl_drop) == [40] * 4
        # per-chunk integer checksums combine to the same digest however the chunks were spread over ranks
        sums = np.array([int(data[a:b].view(np.int32).astype(np.int64).sum()) for a, b in
                         zip(np.cumsum([0] + lens[:-1]), np.cumsum(lens))], dtype=np.int64)
        c_lo = sum(1 for i in range(len(lens)) if sum(lens[:i]) < lo)
        mine_sums = torch.from_numpy(sums[c_lo:c_lo + len(mine)])
        parts = [torch.zeros(3, dtype=torch.int64) for _ in range(world)]
        pad = torch.zeros(3, dtype=torch.int64)
        pad[:len(mine)] = mine_sums
        dist.all_gather(parts, pad)
        counts = [len(shard_chunks(lens, r, world)[2]) for r in range(world)]
        assert combine_chunk_sums(np.concatenate([parts[r][:counts[r]].numpy() for r in range(world)])) == combine_chunk_sums(sums)

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
