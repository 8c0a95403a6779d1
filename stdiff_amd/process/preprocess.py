"""Pre-processing of the reference driver on the device (test-stDiff.py:44-55; process/data.py:79-100), arrays in,
arrays out (the reference works on AnnData through scanpy / sklearn, which this image does not have):

    data_augment(adata, fixed, noise_std)  -> rows [X ; X + noise_std]  (fixed)  or  [X ; X + |N(0, noise_std)|]
    sc.pp.normalize_total(target_sum=1e4)  -> each cell scaled to target_sum (zero-count cells untouched)
    sc.pp.log1p
    scale(adata) = MaxAbsScaler().fit_transform(X.T).T  -> each cell divided by its largest |value|

normalize_total, log1p and scale are one fused per-cell kernel (stdiff_preprocess_rows); `scale` alone is the same
kernel with the first two stages switched off.  Parity: scale() is pinned by the reference's own function run on
sklearn (golden G10, bit-equal); scanpy is not importable here, so normalize_total / log1p are checked against a
float64 numpy restatement of their documented semantics only (parity unpinned for those two stages)."""
import numpy as np
import torch

from .._lib import check, lib, ptr, require_gpu, stream


def normalize_log1p_scale(x, target_sum=1e4, add_const=0.0, out=None):
    """x: fp32 [cells, genes] device tensor (raw counts) -> fp32 in [0, 1] (for non-negative counts), row max = 1."""
    require_gpu(x, out)
    if x.dim() != 2 or x.dtype != torch.float32 or x.stride(1) != 1:
        raise ValueError("x must be fp32 [cells, genes] with contiguous rows")
    if out is None:
        out = torch.empty_like(x)
    check(lib().stdiff_preprocess_rows(ptr(x), x.stride(0), ptr(out), out.stride(0), x.shape[0], x.shape[1],
                                       float(add_const), float(target_sum), stream()))
    return out


def scale(x, out=None):
    """scale() of the reference (process/data.py:79-83: MaxAbsScaler().fit_transform(X.T).T) alone: every cell divided
    by its largest |value| (cells of zeros by 1), fp32 division as sklearn does it -- bit-equal to the reference on
    fp32 input (golden G10).  x: fp32 [cells, genes] device tensor."""
    require_gpu(x, out)
    if x.dim() != 2 or x.dtype != torch.float32 or x.stride(1) != 1:
        raise ValueError("x must be fp32 [cells, genes] with contiguous rows")
    if out is None:
        out = torch.empty_like(x)
    check(lib().stdiff_preprocess_rows(ptr(x), x.stride(0), ptr(out), out.stride(0), x.shape[0], x.shape[1], 0.0, 0.0,
                                       stream()))
    return out


def preprocess_pair(seq_counts, spatial_counts, noise_std=10.0, fixed=True, device="cuda:0", target_sum=1e4,
                    rng=None):
    """The driver's preprocessing block (test-stDiff.py:44-55): returns (data_seq_array [2*Nseq, G], data_spatial_array
    [Nsp, G]) as float32 numpy, ready for diffusion_impute.  fixed=False draws |N(0, noise_std)| with numpy's
    generator `rng` on the host, like the reference (np.random.normal)."""
    dev = torch.device(device)
    seq = torch.as_tensor(np.ascontiguousarray(seq_counts, dtype=np.float32)).to(dev)
    sp = torch.as_tensor(np.ascontiguousarray(spatial_counts, dtype=np.float32)).to(dev)
    n, g = seq.shape
    both = torch.empty((2 * n, g), device=dev, dtype=torch.float32)
    normalize_log1p_scale(seq, target_sum, 0.0, out=both[:n])
    if fixed:
        normalize_log1p_scale(seq, target_sum, float(noise_std), out=both[n:])
    else:
        rng = rng or np.random.default_rng()
        aug = seq + torch.as_tensor(np.abs(rng.normal(0, noise_std, (n, g))).astype(np.float32)).to(dev)
        normalize_log1p_scale(aug, target_sum, 0.0, out=both[n:])
    return both.cpu().numpy(), normalize_log1p_scale(sp, target_sum).cpu().numpy()
