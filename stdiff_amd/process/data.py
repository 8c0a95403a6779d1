"""get_data_loader with the reference signature (process/data.py:64-75): mini-batches of (x, x_cond) rows.
The batch boundaries ARE semantics here: the denoiser attends across the cells of a batch, so sample_stDiff and
normal_train_stDiff take their chunking from this loader (drop_last=False: the last batch may be partial)."""
import numpy as np
import torch
from torch.utils.data import DataLoader, TensorDataset


def get_data_loader(data_ary: np.ndarray,
                    cell_type: np.ndarray,
                    batch_size: int = 512,
                    is_shuffle: bool = True,
                    ):
    data_tensor = torch.from_numpy(np.ascontiguousarray(data_ary, dtype=np.float32))
    cell_type_tensor = torch.from_numpy(np.ascontiguousarray(cell_type, dtype=np.float32))
    dataset = TensorDataset(data_tensor, cell_type_tensor)
    # the reference builds a CUDA generator only because it sets the CUDA default tensor type
    # (test-stDiff.py:18); the shuffle permutation itself is host-side index arithmetic.
    generator = torch.Generator(device='cpu')
    generator.manual_seed(torch.initial_seed() & 0x7FFFFFFFFFFFFFFF)
    return DataLoader(dataset, batch_size=batch_size, shuffle=is_shuffle, drop_last=False, generator=generator)
