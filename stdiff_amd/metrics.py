"""Per-gene accuracy metrics of the reference's evaluation block (test-stDiff.py:203-364) from two HBM passes over
the imputed and measured matrices (csrc/metrics.hip); the closed forms below run on [G]-sized float64 vectors.

Covered: PCC (per-gene Pearson), cosine (north_star), SPCC (per-gene Spearman: average ranks by counting, exact),
RMSE after zscore, SSIM after scale_max, JS after scale_plus.  Not covered: the scanpy clustering scores."""
import numpy as np
import torch

from ._lib import check, lib, ptr, require_gpu, stream


def _chunks(n):
    return int(min(64, max(1, (n + 255) // 256)))


def column_moments(x, y):
    """x, y: fp32 [N, G] device tensors -> dict of float64 numpy [G]: n, sx, sy, sxy, sxx, syy, maxx, maxy."""
    require_gpu(x, y)
    if x.shape != y.shape or x.dim() != 2 or x.dtype != torch.float32 or y.dtype != torch.float32:
        raise ValueError("x and y must be fp32 [N, G] of equal shape")
    if x.stride(1) != 1 or y.stride(1) != 1:
        raise ValueError("rows must be contiguous")
    N, G = x.shape
    ch = _chunks(N)
    part = torch.empty((ch, 7, G), device=x.device, dtype=torch.float64)
    check(lib().stdiff_column_moments(ptr(x), x.stride(0), ptr(y), y.stride(0), ptr(part), ch, N, G, stream()))
    p = part.cpu().numpy()
    m = {"n": float(N)}
    for k, name in enumerate(("sx", "sy", "sxy", "sxx", "syy")):
        acc = np.zeros(G)
        for c in range(ch):                    # slab order: deterministic
            acc += p[c, k]
        m[name] = acc
    m["maxx"], m["maxy"] = p[:, 5].max(axis=0), p[:, 6].max(axis=0)
    return m


def _pearson(m):
    n = m["n"]
    cov = m["sxy"] - m["sx"] * m["sy"] / n
    vx = m["sxx"] - m["sx"] ** 2 / n
    vy = m["syy"] - m["sy"] ** 2 / n
    with np.errstate(divide="ignore", invalid="ignore"):
        return cov / np.sqrt(vx * vy)


def _ssim_scale_max(m):
    """cal_ssim (:203-221) of the columns after scale_max (:224-229): both maxima become 1, so M = 1."""
    n = m["n"]
    ax, ay = 1.0 / m["maxx"], 1.0 / m["maxy"]
    mu1, mu2 = m["sx"] / n * ax, m["sy"] / n * ay
    var1 = (m["sxx"] / n) * ax * ax - mu1 ** 2
    var2 = (m["syy"] / n) * ay * ay - mu2 ** 2
    s1, s2 = np.sqrt(np.maximum(var1, 0)), np.sqrt(np.maximum(var2, 0))
    s12 = (m["sxy"] / n) * ax * ay - mu1 * mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    C3 = C2 / 2
    l12 = (2 * mu1 * mu2 + C1) / (mu1 ** 2 + mu2 ** 2 + C1)
    c12 = (2 * s1 * s2 + C2) / (s1 ** 2 + s2 ** 2 + C2)
    return l12 * c12 * (s12 + C3) / (s1 * s2 + C3)


def spearman(x, y):
    """Per-gene Spearman rho of two fp32 [N, G] device matrices -> float64 numpy [G] (scipy.stats.spearmanr per column;
    a constant column gives nan, as scipy does).  Work grows as N^2 G (20 000 x 1000: ~0.1 s)."""
    from . import ops
    require_gpu(x, y)
    if x.shape != y.shape or x.dim() != 2 or x.dtype != torch.float32 or y.dtype != torch.float32:
        raise ValueError("x and y must be fp32 [N, G] of equal shape")
    N, G = x.shape
    out = np.empty(G)
    blocks = (N + 1023) // 1024
    for g0 in range(0, G, 32768):                        # grid.y limit of one launch
        g1 = min(G, g0 + 32768)
        xt, yt = ops.transpose(x[:, g0:g1]), ops.transpose(y[:, g0:g1])
        part = torch.empty((g1 - g0, blocks, 3), device=x.device, dtype=torch.float64)
        check(lib().stdiff_rank_corr(ptr(xt), ptr(yt), xt.stride(0), ptr(part), blocks, N, g1 - g0, stream()))
        p = part.cpu().numpy()
        acc = np.zeros((g1 - g0, 3))
        for b in range(blocks):                          # block order: deterministic
            acc += p[:, b]
        with np.errstate(divide="ignore", invalid="ignore"):
            out[g0:g1] = acc[:, 2] / np.sqrt(acc[:, 0] * acc[:, 1])
    return out


def _degenerate_column(x, y):
    """SSIM / RMSE / JS of ONE gene (float64 host vectors: x measured, y imputed) whose measured or imputed column is
    constant or all zero.  There the reference's scaling step divides 0 by 0 and its `fillna(1e-20)`
    (test-stDiff.py:285-286,332-333,356-357) decides the value: NaNs of the SCALED column become 1e-20.  The closed
    forms of the device path do not cover that, and such genes are few, so they take this [N]-sized host route."""
    def fill(v):
        v = np.array(v, dtype=np.float64)
        v[np.isnan(v)] = 1e-20
        return v
    out = {}
    with np.errstate(all="ignore"):
        a, b = fill(x / x.max()), fill(y / y.max())                                    # scale_max :223-228
        M = b.max() if a.max() > b.max() else a.max()                                  # :287 (indexing a pair by a bool)
        mu1, mu2 = a.mean(), b.mean()                                                  # cal_ssim :203-220
        s1, s2 = np.sqrt(((a - mu1) ** 2).mean()), np.sqrt(((b - mu2) ** 2).mean())
        s12 = ((a - mu1) * (b - mu2)).mean()
        C1, C2 = (0.01 * M) ** 2, (0.03 * M) ** 2
        C3 = C2 / 2
        out["SSIM"] = ((2 * mu1 * mu2 + C1) / (mu1 ** 2 + mu2 ** 2 + C1)) * ((2 * s1 * s2 + C2) / (s1 ** 2 + s2 ** 2 + C2)) \
            * ((s12 + C3) / (s1 * s2 + C3))
        zx, zy = fill((x - x.mean()) / x.std()), fill((y - y.mean()) / y.std())        # scale_z_score :231-237
        out["RMSE"] = np.sqrt(((zx - zy) ** 2).mean())                                 # :358
        p, q = fill(x / x.sum()), fill(y / y.sum())                                    # scale_plus :240-245
        m = (p + q) / 2                                                                # :334-335

        def kl(pk, qk):                                                                # scipy.stats.entropy(pk, qk)
            pk, qk = pk / pk.sum(), qk / qk.sum()
            return np.where(pk > 0, pk * np.log(pk / qk), 0.0).sum()
        out["JS"] = 0.5 * kl(p, m) + 0.5 * kl(q, m)
    return out


def gene_metrics(imputed, measured, js=True, spcc=True):
    """dict of per-gene float64 numpy vectors: PCC, cosine, SPCC, RMSE (after zscore), SSIM (after scale_max), JS
    (after scale_plus).  imputed / measured: [N, G] numpy or device tensors (raw = measured, impute = imputed in the
    reference's naming)."""
    dev = imputed.device if isinstance(imputed, torch.Tensor) else (
        measured.device if isinstance(measured, torch.Tensor) else torch.device("cuda:0"))
    x = torch.as_tensor(np.asarray(measured, dtype=np.float32) if not isinstance(measured, torch.Tensor) else measured)
    y = torch.as_tensor(np.asarray(imputed, dtype=np.float32) if not isinstance(imputed, torch.Tensor) else imputed)
    x, y = x.float().to(dev).contiguous(), y.float().to(dev).contiguous()
    m = column_moments(x, y)
    pcc = _pearson(m)
    out = {"PCC": pcc}
    with np.errstate(divide="ignore", invalid="ignore"):
        out["cosine"] = m["sxy"] / np.sqrt(m["sxx"] * m["syy"])
        # zscore (ddof 0) of both columns: mean((zx - zy)^2) = 2 - 2 r
        out["RMSE"] = np.sqrt(np.maximum(2.0 - 2.0 * pcc, 0.0))
        out["SSIM"] = _ssim_scale_max(m)
    if spcc:
        out["SPCC"] = spearman(x, y)
    if js:
        N, G = x.shape
        ch = _chunks(N)
        with np.errstate(divide="ignore"):
            ix = torch.from_numpy(1.0 / m["sx"]).to(dev)
            iy = torch.from_numpy(1.0 / m["sy"]).to(dev)
        part = torch.empty((ch, G), device=dev, dtype=torch.float64)
        check(lib().stdiff_column_js(ptr(x), x.stride(0), ptr(y), y.stride(0), ptr(ix), ptr(iy), ptr(part), ch, N, G,
                                     stream()))
        p = part.cpu().numpy()
        acc = np.zeros(G)
        for c in range(ch):
            acc += p[c]
        out["JS"] = acc
    # constant / all-zero columns: the reference's fillna(1e-20) semantics (see _degenerate_column)
    n = m["n"]
    bad = np.flatnonzero((m["sxx"] - m["sx"] ** 2 / n <= 0) | (m["syy"] - m["sy"] ** 2 / n <= 0))
    if bad.size:
        idx = torch.from_numpy(bad).to(dev)
        xc, yc = x[:, idx].double().cpu().numpy(), y[:, idx].double().cpu().numpy()
        for k, g in enumerate(bad):
            d = _degenerate_column(xc[:, k], yc[:, k])
            for name in ("SSIM", "RMSE") + (("JS",) if js else ()):
                out[name][g] = d[name]
    return out
