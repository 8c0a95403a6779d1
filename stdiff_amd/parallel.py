"""Multi-GPU: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm).

Sampling shards naturally: the unit is the reference CHUNK (a DataLoader batch: attention couples the rows inside
it, model/sample.py:9-18), chunks are independent, so each rank owns a contiguous run of chunks and the reverse loop
needs NO collective; in-kernel noise is keyed by the global row index, so results do not depend on the world size.
One all_gather of the [rows/W, G] results at the end (optional).
Training is data parallel: see TrainEngine (flat gradient buffer all-reduced per DiT block, overlapped with backward).
"""
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


def gather_rows(local, total_rows, group=None):
    """all_gather of variable-length row shards [r_i, G] (host numpy) -> [total_rows, G] on every rank."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    parts = [None] * world
    dist.all_gather_object(parts, np.ascontiguousarray(local), group=group)
    out = np.concatenate([p for p in parts if p.shape[0] > 0], axis=0)
    assert out.shape[0] == total_rows, (out.shape, total_rows)
    return out


def sample_stDiff_sharded(model, dataloader, noise_scheduler, mask, gt, num_step=1000, sample_shape=None,
                          sample_intermediate=None, model_pred_type='noise', seed=0, x_T=None, step_noise=None,
                          gather=True, group=None, use_graph=True):
    """sample_stDiff (model/sample.py:22-101) with the spots sharded over the ranks of `group`.
    Every rank passes the SAME dataloader / gt / mask (as in a replicated driver script) and keeps only its chunks.
    Returns the full [N, G] ndarray on every rank (gather=True) or this rank's (row_lo, row_hi, shard)."""
    import torch.distributed as dist
    from .model.sample import _collect_condition
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    eng = model.eval().engine()
    dev = eng.pw.device
    x_cond, lens = _collect_condition(dataloader, dev)
    N, G = x_cond.shape
    lo, hi, my_lens = shard_chunks(lens, rank, world)
    mask_t = torch.as_tensor(np.asarray(mask), dtype=torch.float32).to(dev).contiguous()
    gt_t = torch.as_tensor(np.asarray(gt), dtype=torch.float32)[lo:hi].to(dev).contiguous()
    timesteps = list(range(num_step))[::-1]
    if sample_intermediate:
        timesteps = timesteps[:sample_intermediate]
    if hi > lo:
        with torch.no_grad():
            smp = E.LanedSampler(eng, noise_scheduler, my_lens, x_cond[lo:hi].contiguous(), gt_t, mask_t,
                                 pred_type=model_pred_type, row_offset=lo, seed=seed, use_graph=use_graph,
                                 lanes=1 if step_noise is not None else None)
            smp.init(None if x_T is None else torch.as_tensor(x_T, dtype=torch.float32)[lo:hi].to(dev).contiguous())
            z = None if step_noise is None else torch.as_tensor(step_noise, dtype=torch.float32)[:, lo:hi].to(dev).contiguous()
            smp.run(timesteps, z)
            local = smp.x.detach().cpu().numpy()
    else:
        local = np.zeros((0, G), dtype=np.float32)
    if not gather:
        return lo, hi, local
    return gather_rows(local, N, group)
