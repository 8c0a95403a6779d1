"""Multi-GPU: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm).

Sampling shards naturally: the unit is the reference CHUNK (a DataLoader batch: attention couples the rows inside
it, model/sample.py:9-18), chunks are independent, so each rank owns a contiguous run of chunks and the reverse loop
needs NO collective; in-kernel noise is keyed by the global row index, so results do not depend on the world size.
A rank uploads ONLY its rows; the optional final gather is one device all_gather of the padded [rows/W, G] shards.
`ShardedSampler` is the object behind `sample_stDiff_sharded` AND behind `bench.py --gpus N`: what is measured is what
a user calls.
Training is data parallel: see TrainEngine (flat gradient buffer all-reduced per DiT block, overlapped with backward).
"""
import hashlib

import numpy as np
import torch

from . import engine as E


def shard_chunks(chunk_lens, rank, world):
    """Contiguous, chunk-aligned partition: returns (row_lo, row_hi, chunk_lens_of_this_rank)."""
    n = len(chunk_lens)
    base, extra = divmod(n, world)
    lo_c = rank * base + min(rank, extra)
    hi_c = lo_c + base + (1 if rank < extra else 0)
    cu = np.concatenate([[0], np.cumsum(chunk_lens)]).astype(np.int64)
    return int(cu[lo_c]), int(cu[hi_c]), [int(c) for c in chunk_lens[lo_c:hi_c]]


def _dist(group=None):
    """(dist module or None, rank, world) of `group` -- (None, 0, 1) when no process group is initialised."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return None, 0, 1
    return dist, dist.get_rank(group), dist.get_world_size(group)


def gather_rows(local, total_rows, group=None):
    """all_gather of variable-length row shards [r_i, G] -> [total_rows, G] on every rank.

    `local` is a torch tensor (device or host) or a numpy array; the result has the same kind.  The shards are padded
    to the longest one and exchanged with ONE all_gather_into_tensor on the tensor's own device (RCCL for device
    tensors under backend "nccl"; host tensors under gloo) -- no pickling, no per-rank host staging of the whole
    matrix."""
    dist, rank, world = _dist(group)
    as_numpy = isinstance(local, np.ndarray)
    t = torch.from_numpy(np.ascontiguousarray(local)) if as_numpy else local.contiguous()
    if dist is None or world == 1:
        assert t.shape[0] == total_rows, (tuple(t.shape), total_rows)
        return local
    if dist.get_backend(group) == "gloo" and t.is_cuda:            # rehearsal / tests: gloo moves host memory
        out = gather_rows(t.cpu(), total_rows, group)
        return out.numpy() if as_numpy else out.to(t.device)
    G = t.shape[1]
    rows = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
    all_rows = torch.empty(world, dtype=torch.int64, device=t.device)
    dist.all_gather_into_tensor(all_rows, rows, group=group)
    all_rows = [int(v) for v in all_rows.cpu()]
    assert sum(all_rows) == total_rows, (all_rows, total_rows)
    rmax = max(all_rows)
    if t.shape[0] == rmax:
        padded = t
    else:
        padded = torch.zeros((rmax, G), dtype=t.dtype, device=t.device)
        padded[:t.shape[0]].copy_(t)
    buf = torch.empty((world * rmax, G), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(buf, padded, group=group)
    if all(r == rmax for r in all_rows):
        out = buf
    else:
        out = torch.cat([buf[i * rmax:i * rmax + r] for i, r in enumerate(all_rows)], dim=0)
    return out.numpy() if as_numpy else out


def loader_chunk_lens(dataloader):
    """Chunk (mini-batch) lengths of a loader in loader order WITHOUT moving any data where the loader says so itself:
    an unshuffled DeviceLoader, or a torch DataLoader over a sized dataset with its default sequential batch sampler.
    Anything else is iterated once on the host (lengths only)."""
    from .process.data import DeviceLoader
    if isinstance(dataloader, DeviceLoader):
        if dataloader.is_shuffle or dataloader.world != 1:
            raise ValueError("sharded sampling needs the unshuffled, unsharded loader of sample_stDiff (is_shuffle=False)")
        return list(dataloader.lens)
    from torch.utils.data import DataLoader, SequentialSampler
    if isinstance(dataloader, DataLoader) and dataloader.batch_size and isinstance(dataloader.sampler, SequentialSampler):
        n, b = len(dataloader.dataset), int(dataloader.batch_size)
        lens = [min(b, n - i) for i in range(0, n, b)]
        if dataloader.drop_last and lens and lens[-1] < b:
            lens.pop()
        return lens
    return [int(x_cond.shape[0]) for _, x_cond in dataloader]


def loader_condition_rows(dataloader, lo, hi, device):
    """Condition rows [lo, hi) of the loader (second element of its batches, sample.py:9) as a device tensor; only
    these rows are uploaded."""
    from .process.data import DeviceLoader
    if isinstance(dataloader, DeviceLoader):
        x = dataloader.x_cond[lo:hi]
        return x if x.device == torch.empty(0, device=device).device else x.to(device)
    from torch.utils.data import DataLoader, SequentialSampler, TensorDataset
    if isinstance(dataloader, DataLoader) and isinstance(dataloader.sampler, SequentialSampler) \
            and isinstance(dataloader.dataset, TensorDataset) and len(dataloader.dataset.tensors) >= 2:
        return dataloader.dataset.tensors[1][lo:hi].float().to(device).contiguous()
    parts, r = [], 0
    for _, x_cond in dataloader:
        n = int(x_cond.shape[0])
        a, b = max(lo, r), min(hi, r + n)
        if a < b:
            parts.append(x_cond[a - r:b - r].float().to(device))
        r += n
        if r >= hi:
            break
    if not parts:
        return torch.empty((0, 0), device=device)
    return torch.cat(parts, dim=0).contiguous()


def combine_chunk_sums(chunk_sums):
    """sha256 over the per-chunk int64 checksums of ALL chunks in global chunk order (hex digest)."""
    a = np.ascontiguousarray(np.asarray(chunk_sums, dtype=np.int64))
    return hashlib.sha256(a.tobytes()).hexdigest()


class ShardedSampler:
    """This rank's part of a spot-sharded sample_stDiff run.

    all_lens: chunk lengths of the WHOLE matrix in loader order; `fetch(lo, hi)` returns this rank's condition rows and
    known-gene rows as fp32 device tensors [hi-lo, G] (so only the shard ever reaches this GPU).  Without a process
    group it is the single-process sampler (rank 0 of 1)."""

    def __init__(self, model, noise_scheduler, mask, all_lens, fetch, *, group=None, model_pred_type='noise', seed=0,
                 use_graph=True, lanes=None):
        self.dist, self.rank, self.world = _dist(group)
        self.group = group
        self.all_lens = [int(c) for c in all_lens]
        self.N = int(sum(self.all_lens))
        eng = model.eval().engine()
        self.eng, self.dev = eng, eng.pw.device
        self.lo, self.hi, self.lens = shard_chunks(self.all_lens, self.rank, self.world)
        self.mask = torch.as_tensor(np.asarray(mask), dtype=torch.float32).to(self.dev).contiguous() \
            if not torch.is_tensor(mask) else mask.to(self.dev).float().contiguous()
        self.G = int(self.mask.shape[0])
        self.smp = self.gt = None
        if self.hi > self.lo:
            x_cond, gt = fetch(self.lo, self.hi)
            self.gt = gt
            if x_cond.shape != (self.hi - self.lo, self.G) or gt.shape != x_cond.shape:
                raise ValueError(f"fetch({self.lo}, {self.hi}) returned {tuple(x_cond.shape)} / {tuple(gt.shape)}")
            with torch.no_grad():
                self.smp = E.LanedSampler(eng, noise_scheduler, self.lens, x_cond.contiguous(), gt.contiguous(), self.mask,
                                          pred_type=model_pred_type, row_offset=self.lo, seed=seed, use_graph=use_graph,
                                          lanes=lanes)

    def init(self, x_T=None):
        if self.smp is not None:
            with torch.no_grad():
                self.smp.init(x_T)

    def run(self, timesteps, z=None):
        if self.smp is not None:
            with torch.no_grad():
                self.smp.run(timesteps, z)

    def run_more(self, n):
        if self.smp is not None:
            with torch.no_grad():
                self.smp.run_more(n)

    @property
    def x(self):
        """This rank's rows [lo, hi) of the state, fp32 on the device."""
        if self.smp is None:
            return torch.empty((0, self.G), device=self.dev, dtype=torch.float32)
        return self.smp.x

    def gather(self):
        """The whole [N, G] state on every rank (device tensor)."""
        return gather_rows(self.x, self.N, self.group)

    def chunk_checksums(self):
        """int64 [n_chunks of this rank]: for every chunk the sum of the fp32 BIT PATTERNS of its rows (as integers:
        exact and independent of the summation order, unlike a floating-point sum)."""
        out, r = [], 0
        x = self.x
        blk = 64                                        # chunks per pass: bounds the int64 temporary
        for c0 in range(0, len(self.lens), blk):
            lens = self.lens[c0:c0 + blk]
            rows = sum(lens)
            bits = x[r:r + rows].view(torch.int32).to(torch.int64).sum(dim=1)          # [rows]
            cu = np.concatenate([[0], np.cumsum(lens)])
            if len(set(lens)) == 1:
                out.append(bits.view(len(lens), lens[0]).sum(dim=1))
            else:
                out.append(torch.stack([bits[cu[i]:cu[i + 1]].sum() for i in range(len(lens))]))
            r += rows
        return torch.cat(out) if out else torch.empty(0, dtype=torch.int64, device=self.dev)

    def checksum(self):
        """(hex digest over the per-chunk checksums of all ranks in global chunk order, float64 sum of the whole state).
        Equal for every world size by construction (results do not depend on the sharding); collective call."""
        mine = self.chunk_checksums()
        total = self.x.double().sum().reshape(1) if self.x.numel() else torch.zeros(1, dtype=torch.float64, device=self.dev)
        if self.dist is not None and self.world > 1:
            n_all = len(self.all_lens)
            counts = [len(shard_chunks(self.all_lens, r, self.world)[2]) for r in range(self.world)]
            cmax = max(counts)
            # int64 checksums travel as two exact float64 halves next to the float64 total (one collective, and gloo
            # as well as RCCL move float64)
            both = torch.zeros((2, cmax + 1), dtype=torch.float64, device=self.dev)
            both[0, :len(mine)] = (mine >> 32).double()
            both[1, :len(mine)] = (mine & 0xFFFFFFFF).double()
            both[0, cmax] = total[0]
            buf = torch.empty(self.world * 2 * (cmax + 1), dtype=torch.float64, device=self.dev)      # flat: gloo insists
            if self.dist.get_backend(self.group) == "gloo":
                hb = buf.cpu()
                self.dist.all_gather_into_tensor(hb, both.reshape(-1).cpu(), group=self.group)
                buf = hb
            else:
                self.dist.all_gather_into_tensor(buf, both.reshape(-1), group=self.group)
            buf = buf.cpu().view(self.world, 2, cmax + 1)
            parts = []
            for r in range(self.world):
                hi32 = buf[r, 0, :counts[r]].to(torch.int64)
                lo32 = buf[r, 1, :counts[r]].to(torch.int64)
                parts.append(((hi32 << 32) | lo32).numpy())
            sums = np.concatenate(parts)
            assert sums.shape[0] == n_all
            tot = float(buf[:, 0, cmax].sum())
        else:
            sums, tot = mine.cpu().numpy(), float(total.item())
        return combine_chunk_sums(sums), tot


def sample_stDiff_sharded(model, dataloader, noise_scheduler, mask, gt, num_step=1000, sample_shape=None,
                          sample_intermediate=None, model_pred_type='noise', seed=0, x_T=None, step_noise=None,
                          gather=True, group=None, use_graph=True, lanes=None):
    """sample_stDiff (model/sample.py:22-101) with the spots sharded over the ranks of `group`.
    Every rank passes the SAME dataloader / gt / mask (as in a replicated driver script); each rank uploads and keeps
    only its chunks' rows.  Returns the full [N, G] ndarray on every rank (gather=True: one device all_gather of the
    padded shards, then one D2H) or this rank's (row_lo, row_hi, shard ndarray).  `seed` must be the same on every rank
    (it keys the noise by global row)."""
    all_lens = loader_chunk_lens(dataloader)
    gt_np = np.asarray(gt) if not torch.is_tensor(gt) else gt
    N = int(sum(all_lens))
    if sample_shape is not None and (int(sample_shape[0]) != N or tuple(gt_np.shape) != tuple(int(v) for v in sample_shape)):
        raise ValueError(f"sample_shape {tuple(sample_shape)} does not match dataloader rows {N} / gt {tuple(gt_np.shape)}")
    dev = model.eval().engine().pw.device

    def rows(a, lo, hi):
        t = a[lo:hi] if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a[lo:hi], dtype=np.float32))
        return t.float().to(dev).contiguous()

    def fetch(lo, hi):
        return loader_condition_rows(dataloader, lo, hi, dev), rows(gt_np, lo, hi)

    ss = ShardedSampler(model, noise_scheduler, mask, all_lens, fetch, group=group, model_pred_type=model_pred_type,
                        seed=seed, use_graph=use_graph, lanes=1 if step_noise is not None else lanes)
    timesteps = list(range(num_step))[::-1]
    if sample_intermediate:
        timesteps = timesteps[:sample_intermediate]
    lo, hi = ss.lo, ss.hi
    ss.init(None if x_T is None else rows(x_T, lo, hi))
    z = None
    if step_noise is not None and hi > lo:
        z = (step_noise[:, lo:hi] if torch.is_tensor(step_noise)
             else torch.from_numpy(np.ascontiguousarray(step_noise[:, lo:hi], dtype=np.float32))).float().to(dev).contiguous()
    ss.run(timesteps, z)
    if not gather:
        return lo, hi, ss.x.detach().cpu().numpy()
    return ss.gather().detach().cpu().numpy()
