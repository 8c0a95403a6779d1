"""get_data_loader with the reference signature (process/data.py:64-75): mini-batches of (x, x_cond) rows.
The batch boundaries ARE semantics here: the denoiser attends across the cells of a batch, so sample_stDiff and
normal_train_stDiff take their chunking from this loader (drop_last=False: the last batch may be partial).

Two loaders with the same batch sequence:
  * the host `DataLoader` of the reference (default), and
  * `DeviceLoader` (`device=` given): both matrices live in HBM once; an unshuffled pass yields VIEWS, a shuffled pass
    draws the epoch permutation exactly as the reference's RandomSampler does (`torch.randperm(n, generator=g)` on a
    CPU generator) and gathers each batch with one HIP kernel.  The reference re-uploads every batch on every use
    (stDiff_train.py:60-61, sample.py:10: once per reverse step)."""
import numpy as np
import torch
from torch.utils.data import DataLoader, TensorDataset


def _host_generator():
    # the reference builds a CUDA generator only because it sets the CUDA default tensor type
    # (test-stDiff.py:18); the shuffle permutation itself is host-side index arithmetic.
    generator = torch.Generator(device='cpu')
    generator.manual_seed(torch.initial_seed() & 0x7FFFFFFFFFFFFFFF)
    return generator


class DeviceLoader:
    """Iterable of (x, x_cond) fp32 device batches over device-resident matrices.

    rank / world: chunk-aligned sharding for data-parallel training: rank r takes batches r, r + world, ... of every
    epoch; an incomplete last round is dropped so that all ranks run the same number of iterations (their gradient
    all-reduces must pair up).  world = 1 keeps every batch, like the reference."""

    def __init__(self, data_ary, cell_type, batch_size=512, is_shuffle=True, device='cuda:0', rank=0, world=1,
                 generator=None):
        from .. import _lib
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("stdiff_amd runs on MI355X only: DeviceLoader needs a cuda device (no CPU fallback); "
                               "use get_data_loader(...) without device= for the host loader")
        _lib.lib()                                               # fail early if the HIP extension is missing
        self.x = torch.as_tensor(np.ascontiguousarray(data_ary, dtype=np.float32)).to(dev).contiguous()
        self.x_cond = torch.as_tensor(np.ascontiguousarray(cell_type, dtype=np.float32)).to(dev).contiguous()
        if self.x.shape != self.x_cond.shape or self.x.dim() != 2:
            raise ValueError(f"data {tuple(self.x.shape)} and condition {tuple(self.x_cond.shape)} must be equal [N, G]")
        if batch_size <= 0 or not (0 <= rank < world):
            raise ValueError("batch_size > 0 and 0 <= rank < world required")
        self.batch_size, self.is_shuffle, self.device = int(batch_size), bool(is_shuffle), dev
        self.rank, self.world = int(rank), int(world)
        self.generator = generator if generator is not None else _host_generator()
        n = self.x.shape[0]
        self.bounds = [(i, min(i + self.batch_size, n)) for i in range(0, n, self.batch_size)]
        if self.world > 1:
            full = len(self.bounds) // self.world * self.world
            self.bounds = self.bounds[:full][self.rank::self.world]
        self.lens = [b - a for a, b in self.bounds]

    def __len__(self):
        return len(self.bounds)

    def __iter__(self):
        if not self.is_shuffle:
            for a, b in self.bounds:
                yield self.x[a:b], self.x_cond[a:b]
            return
        from .. import ops
        # the generator is consumed exactly as DataLoader + RandomSampler consume it per epoch (torch 2.x: the iterator's
        # base seed, the epoch permutation, and one discarded permutation for the empty `num_samples % n` tail), so the
        # same seed gives the same batches as the host loader (tests/test_gpu_model.py)
        n = self.x.shape[0]
        torch.empty((), dtype=torch.int64).random_(generator=self.generator)
        perm = torch.randperm(n, generator=self.generator)
        torch.randperm(n, generator=self.generator)
        perm = perm.to(self.device, non_blocking=True)
        for a, b in self.bounds:
            yield ops.gather_rows2(self.x, self.x_cond, perm[a:b])


def get_data_loader(data_ary: np.ndarray,
                    cell_type: np.ndarray,
                    batch_size: int = 512,
                    is_shuffle: bool = True,
                    *,
                    device=None,
                    rank: int = 0,
                    world: int = 1,
                    ):
    """Reference call: a host DataLoader.  Keyword-only extras: `device` returns a DeviceLoader (same batches, data
    resident in HBM); `rank`/`world` shard its batches for data-parallel training."""
    if device is not None:
        return DeviceLoader(data_ary, cell_type, batch_size, is_shuffle, device, rank, world)
    if world != 1:
        raise ValueError("rank/world sharding is a DeviceLoader feature: pass device=")
    data_tensor = torch.from_numpy(np.ascontiguousarray(data_ary, dtype=np.float32))
    cell_type_tensor = torch.from_numpy(np.ascontiguousarray(cell_type, dtype=np.float32))
    dataset = TensorDataset(data_tensor, cell_type_tensor)
    return DataLoader(dataset, batch_size=batch_size, shuffle=is_shuffle, drop_last=False, generator=_host_generator())
