"""Synthetic expression matrices and gene hold-out masks shaped like the reference driver's inputs
(test-stDiff.py:47-55,73,92-108).  Host-side numpy (plus two on-device generators for bench.py); used by bench.py, the tests and the golden generator."""
import numpy as np


def synth_expression(n, g, seed):
    """Poisson-LogNormal counts -> normalize_total(1e4) -> log1p -> per-row max-abs scale: fp32 in [0, 1]."""
    rng = np.random.default_rng(seed)
    lam = rng.lognormal(0.0, 1.0, size=g)
    s = rng.lognormal(0.0, 0.5, size=n)
    out = np.empty((n, g), dtype=np.float32)
    blk = 8192
    for i in range(0, n, blk):
        k = rng.poisson(lam[None, :] * s[i:i + blk, None]).astype(np.float64)
        tot = k.sum(1, keepdims=True)
        tot[tot == 0] = 1.0
        k = np.log1p(k / tot * 1e4)
        mx = np.abs(k).max(1, keepdims=True)
        mx[mx == 0] = 1.0
        out[i:i + blk] = (k / mx).astype(np.float32)
    return out


def kfold_mask(g, fold=0, n_splits=5, seed=0):
    """mask of test-stDiff.py:73,92-96 without sklearn: KFold(shuffle=True, random_state) permutes arange(g) with
    RandomState(seed) and cuts it into n_splits consecutive folds; test genes -> 0, known genes -> 1."""
    idx = np.arange(g)
    np.random.RandomState(seed).shuffle(idx)
    sizes = np.full(n_splits, g // n_splits, dtype=int)
    sizes[: g % n_splits] += 1
    start = int(sizes[:fold].sum())
    test = idx[start:start + sizes[fold]]
    m = np.ones((g,), dtype=np.float32)
    m[test] = 0
    return m


def synth_expression_device(n, g, seed, device, block=65536):
    """synth_expression drawn ON THE DEVICE (torch's Poisson sampler) for matrices too large to make on the host in
    reasonable time (BASELINE configs[3]: 1 M spots x 1024 genes = 4 GB): the same Poisson-LogNormal model and the same
    normalize_total -> log1p -> per-row max-abs pipeline, fp32 in [0, 1].  Input synthesis for bench.py only; the
    values differ from the host generator's (another RNG)."""
    import torch
    rng = np.random.default_rng(seed)
    lam = torch.from_numpy(rng.lognormal(0.0, 1.0, size=g).astype(np.float32)).to(device)
    s = torch.from_numpy(rng.lognormal(0.0, 0.5, size=n).astype(np.float32)).to(device)
    gen = torch.Generator(device=device)
    gen.manual_seed(int(seed))
    out = torch.empty((n, g), device=device, dtype=torch.float32)
    for i in range(0, n, block):
        k = torch.poisson(lam[None, :] * s[i:i + block, None], generator=gen)
        tot = k.sum(1, keepdim=True)
        tot[tot == 0] = 1.0
        k = torch.log1p(k / tot * 1e4)
        mx = k.abs().amax(1, keepdim=True)
        mx[mx == 0] = 1.0
        out[i:i + block] = k / mx
    return out


def synth_expression_device_rows(n_total, g, seed, device, lo, hi, chunk=512):
    """Rows [lo, hi) of ONE n_total x g synthetic matrix, drawn on the device so that a row's values depend on its
    GLOBAL index only -- not on which rank holds it or how many ranks there are (lo and hi must be multiples of `chunk`,
    or n_total): every `chunk` rows are drawn from a generator seeded with (seed, global chunk index), on a tensor whose
    shape that chunk has in every sharding.  bench.py's input at any --gpus N is therefore the same matrix, which is
    what lets its shard checksum be compared across world sizes."""
    import torch
    assert (lo % chunk == 0) and (hi % chunk == 0 or hi == n_total) and 0 <= lo <= hi <= n_total
    rng = np.random.default_rng(seed)
    lam = torch.from_numpy(rng.lognormal(0.0, 1.0, size=g).astype(np.float32)).to(device)
    s = torch.from_numpy(rng.lognormal(0.0, 0.5, size=n_total).astype(np.float32)[lo:hi].copy()).to(device)
    gen = torch.Generator(device=device)
    out = torch.empty((hi - lo, g), device=device, dtype=torch.float32)
    for i in range(lo, hi, chunk):
        j = min(i + chunk, hi)
        gen.manual_seed((int(seed) << 32) + i // chunk)
        k = torch.poisson(lam[None, :] * s[i - lo:j - lo, None], generator=gen)
        tot = k.sum(1, keepdim=True)
        tot[tot == 0] = 1.0
        k = torch.log1p(k / tot * 1e4)
        mx = k.abs().amax(1, keepdim=True)
        mx[mx == 0] = 1.0
        out[i - lo:j - lo] = k / mx
    return out
