"""CPU oracle for the stDiff hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain eager-PyTorch fp32 restatement (own code, same op order) of
the reference's denoiser / scheduler / sampler / train step.  It is the parity
checker and the timed CPU baseline ("port"); it is NOT part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it.  Nothing under ``stdiff_amd/`` imports it.

Pinned: ``tools/make_goldens.py`` runs the real reference (imported from
``/root/reference`` in the build container) side by side with this file and
writes ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` re-checks this
file against those vectors everywhere.

Citations are ``file:line`` into the reference repo (fdu-wangfeilab/stDiff).
Parameters are a flat ``dict[str, Tensor]`` with the reference state-dict keys
(model/stDiff_model.py:176-196).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------
# NoiseScheduler tables  (model/stDiff_scheduler.py:7-66)
# --------------------------------------------------------------------------


def cosine_betas(T, max_beta=0.999):
    """model/stDiff_scheduler.py:7-24, alpha_bar lambda at :43-44 (float64 python)."""
    def alpha_bar(t):
        return math.cos((t + 0.008) / 1.008 * math.pi / 2) ** 2
    out = []
    for i in range(T):
        t1 = i / T
        t2 = (i + 1) / T
        out.append(min(1 - alpha_bar(t2) / alpha_bar(t1), max_beta))
    return np.array(out)


TABLE_NAMES = (
    "betas", "alphas", "alphas_cumprod", "alphas_cumprod_prev",
    "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod",
    "sqrt_inv_alphas_cumprod", "sqrt_inv_alphas_cumprod_minus_one",
    "posterior_mean_coef1", "posterior_mean_coef2",
)


def make_tables(T=1000, beta_start=0.0001, beta_end=0.02, schedule="linear"):
    """All fp32 [T] CPU tables of NoiseScheduler.__init__ (stDiff_scheduler.py:28-66)."""
    if schedule == "linear":
        betas = torch.linspace(beta_start, beta_end, T, dtype=torch.float32)
    elif schedule == "quadratic":
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, T, dtype=torch.float32) ** 2
    elif schedule == "cosine":
        betas = torch.from_numpy(cosine_betas(T).astype(np.float32))
    else:
        raise ValueError(schedule)
    tb = {"betas": betas}
    tb["alphas"] = 1.0 - betas
    tb["alphas_cumprod"] = torch.cumprod(tb["alphas"], dim=0)
    tb["alphas_cumprod_prev"] = F.pad(tb["alphas_cumprod"][:-1], (1, 0), value=1.0)
    tb["sqrt_alphas_cumprod"] = tb["alphas_cumprod"] ** 0.5
    tb["sqrt_one_minus_alphas_cumprod"] = (1 - tb["alphas_cumprod"]) ** 0.5
    tb["sqrt_inv_alphas_cumprod"] = torch.sqrt(1 / tb["alphas_cumprod"])
    tb["sqrt_inv_alphas_cumprod_minus_one"] = torch.sqrt(1 / tb["alphas_cumprod"] - 1)
    tb["posterior_mean_coef1"] = betas * torch.sqrt(tb["alphas_cumprod_prev"]) / (1.0 - tb["alphas_cumprod"])
    tb["posterior_mean_coef2"] = (1.0 - tb["alphas_cumprod_prev"]) * torch.sqrt(tb["alphas"]) / (1.0 - tb["alphas_cumprod"])
    return tb


def variance(tb, t):
    """get_variance (stDiff_scheduler.py:92-100): 0 at t==0 else clipped posterior var."""
    t = int(t)
    if t == 0:
        return torch.tensor(0.0)
    v = tb["betas"][t] * (1.0 - tb["alphas_cumprod_prev"][t]) / (1.0 - tb["alphas_cumprod"][t])
    return v.clip(1e-20)


def add_noise(tb, x0, noise, timesteps):
    """stDiff_scheduler.py:141-148."""
    s1 = tb["sqrt_alphas_cumprod"][timesteps].reshape(-1, 1)
    s2 = tb["sqrt_one_minus_alphas_cumprod"][timesteps].reshape(-1, 1)
    return s1 * x0 + s2 * noise


def reconstruct_x0(tb, x_t, t, eps):
    """stDiff_scheduler.py:68-78 (clamp to [-1, 1])."""
    s1 = tb["sqrt_inv_alphas_cumprod"][t].reshape(-1, 1)
    s2 = tb["sqrt_inv_alphas_cumprod_minus_one"][t].reshape(-1, 1)
    return torch.clamp(s1 * x_t - s2 * eps, min=-1, max=1)


def q_posterior(tb, x0, x_t, t):
    """stDiff_scheduler.py:80-90."""
    s1 = tb["posterior_mean_coef1"][t].reshape(-1, 1)
    s2 = tb["posterior_mean_coef2"][t].reshape(-1, 1)
    return s1 * x0 + s2 * x_t


def step(tb, model_output, t, sample, z=None, pred_type="noise"):
    """stDiff_scheduler.py:103-139. ``z`` is the injected randn_like draw (needed iff t>0)."""
    t = int(t)
    if pred_type == "noise":
        x0 = reconstruct_x0(tb, sample, t, model_output)
    elif pred_type == "x_start":
        x0 = model_output
    else:
        raise NotImplementedError()
    prev = q_posterior(tb, x0, sample, t)
    if t > 0:
        prev = prev + (variance(tb, t) ** 0.5) * z
    return prev, x0


# --------------------------------------------------------------------------
# DiT_stDiff forward  (model/stDiff_model.py)
# --------------------------------------------------------------------------

def timestep_embedding(t, dim=256, max_period=10000):
    """stDiff_model.py:61-80."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


def _lin(p, name, x):
    return F.linear(x, p[name + ".weight"], p[name + ".bias"])


def _ln(x):
    # nn.LayerNorm(H, elementwise_affine=False, eps=1e-6)  (stDiff_model.py:96,99,126)
    return F.layer_norm(x, (x.shape[-1],), eps=1e-6)


def attention(p, pre, x, num_heads):
    """Attention2.forward (stDiff_model.py:23-37): attends ACROSS the rows of x."""
    C, H = x.shape
    d = H // num_heads
    qkv = _lin(p, pre + ".qkv", x).reshape(C, 3, num_heads, d).permute(1, 2, 0, 3)
    q, k, v = qkv[0], qkv[1], qkv[2]                      # [h, C, d]
    att = (q @ k.transpose(-2, -1)) * (d ** -0.5)
    att = att.softmax(dim=-1)
    o = (att @ v).transpose(0, 1).reshape(C, H)
    return _lin(p, pre + ".proj", o)


def time_embed(p, t):
    """TimestepEmbedder.forward (stDiff_model.py:82-85)."""
    e = timestep_embedding(t)
    e = _lin(p, "time_emb.mlp.0", e)
    return _lin(p, "time_emb.mlp.2", F.silu(e))


def dit_forward(p, x, t, y, num_heads, depth, taps=None):
    """DiT_stDiff.forward (stDiff_model.py:231-243) on ONE chunk of rows."""
    x = x.float()
    c = time_embed(p, t) + _lin(p, "cond_layer.0", y)
    h = _lin(p, "in_layer.0", x)
    if taps is not None:
        taps["c"] = c
        taps["h_in"] = h
    sc = F.silu(c)
    for k in range(depth):
        pre = f"blks.{k}"
        mod = F.linear(sc, p[pre + ".adaLN_modulation.1.weight"], p[pre + ".adaLN_modulation.1.bias"])
        sh_a, sc_a, g_a, sh_m, sc_m, g_m = mod.chunk(6, dim=1)
        h = h + g_a * attention(p, pre + ".attn", _ln(h) * (1 + sc_a) + sh_a, num_heads)
        m = _ln(h) * (1 + sc_m) + sh_m
        m = _lin(p, pre + ".mlp.fc2", F.gelu(_lin(p, pre + ".mlp.fc1", m)))
        h = h + g_m * m
        if taps is not None:
            taps[f"h_blk{k}"] = h
    mod = F.linear(sc, p["out_layer.adaLN_modulation.1.weight"], p["out_layer.adaLN_modulation.1.bias"])
    sh, scl = mod.chunk(2, dim=1)
    return _lin(p, "out_layer.linear", _ln(h) * (1 + scl) + sh)


def chunk_bounds(n, batch):
    """DataLoader(batch_size, shuffle=False, drop_last=False) row ranges (process/data.py:74-75)."""
    return [(i, min(i + batch, n)) for i in range(0, n, batch)]


def denoise_chunked(p, x_t, t, y, batch, num_heads, depth):
    """model_sample_stDiff (model/sample.py:6-20): per-chunk forward, concatenated."""
    outs = []
    for a, b in chunk_bounds(x_t.shape[0], batch):
        tt = torch.full((b - a,), int(t), dtype=torch.long)
        outs.append(dit_forward(p, x_t[a:b], tt, y[a:b], num_heads, depth))
    return torch.cat(outs, dim=0)


def sample(p, tb, gt, mask, batch, num_heads, depth, x_T, z_fn, num_step,
           sample_intermediate=None, pred_type="noise", record=None, teacher=None):
    """sample_stDiff (model/sample.py:22-101).

    ``z_fn(t)`` returns the injected noise for step t (called only for t>0, in loop
    order).  ``record`` (dict) receives per-step eps / x_t.  ``teacher`` (dict
    t->x_t) overrides the state fed to the denoiser (teacher forcing).
    """
    mask = torch.as_tensor(mask, dtype=torch.float32)
    gt = torch.as_tensor(gt, dtype=torch.float32)
    x_t = x_T * (1 - mask) + gt * mask
    ts = list(range(num_step))[::-1]
    if sample_intermediate:
        ts = ts[:sample_intermediate]
    with torch.no_grad():
        for t in ts:
            if teacher is not None:
                x_t = teacher[t]
            eps = denoise_chunked(p, x_t, t, gt, batch, num_heads, depth)
            z = z_fn(t) if t > 0 else None
            x_prev, _ = step(tb, eps, t, x_t, z, pred_type)
            x_prev = x_prev * (1.0 - mask) + mask * gt
            if record is not None:
                record.setdefault("eps", {})[t] = eps
                record.setdefault("x_in", {})[t] = x_t
                record.setdefault("x_out", {})[t] = x_prev
            x_t = x_prev
    return x_t


# --------------------------------------------------------------------------
# Train step  (model/stDiff_train.py:56-81)
# --------------------------------------------------------------------------

def trainable_names(p):
    """Every parameter that receives a gradient: all but condi_emb (unused in forward,
    stDiff_model.py:184,236 -> grad None -> skipped by clip and AdamW)."""
    return [k for k in p if k != "condi_emb.weight"]


def loss_and_grads(p, tb, x, x_cond, noise, timesteps, mask, num_heads, depth):
    """Forward + masked MSE + backward of one batch (stDiff_train.py:62-77)."""
    mask = torch.as_tensor(mask, dtype=torch.float32)
    names = trainable_names(p)
    leaves = {k: p[k].detach().clone().requires_grad_(True) for k in names}
    q = dict(p)
    q.update(leaves)
    x_t = add_noise(tb, x, noise, timesteps)
    x_noisy = x_t * (1 - mask) + x * mask
    pred = dit_forward(q, x_noisy, timesteps, x_cond, num_heads, depth)
    loss = F.mse_loss(noise * (1 - mask), pred * (1 - mask))      # mean over C*G
    grads = torch.autograd.grad(loss, [leaves[k] for k in names])
    return loss.detach(), dict(zip(names, grads)), pred.detach()


def clip_grads_(grads, max_norm=1.0):
    """nn.utils.clip_grad_norm_(params, 1.0) (stDiff_train.py:78): returns the pre-clip norm."""
    total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g) for g in grads.values()]))
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads.values():
        g.mul_(coef)
    return total


def adamw_step_(p, grads, state, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
    """torch.optim.AdamW(lr, weight_decay=0) update (stDiff_train.py:47,79)."""
    state["step"] = state.get("step", 0) + 1
    k = state["step"]
    b1, b2 = betas
    bc1 = 1 - b1 ** k
    bc2 = 1 - b2 ** k
    for n, g in grads.items():
        m = state.setdefault("m", {}).setdefault(n, torch.zeros_like(g))
        v = state.setdefault("v", {}).setdefault(n, torch.zeros_like(g))
        w = p[n]
        w.mul_(1 - lr * weight_decay)
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
        w.addcdiv_(m, denom, value=-lr / bc1)


def train_step(p, tb, state, x, x_cond, noise, timesteps, mask, lr, num_heads, depth):
    """One reference iteration: loss, backward, clip(1.0), AdamW. Mutates p/state."""
    loss, grads, _ = loss_and_grads(p, tb, x, x_cond, noise, timesteps, mask, num_heads, depth)
    raw = {k: g.clone() for k, g in grads.items()}
    gnorm = clip_grads_(grads, 1.0)
    adamw_step_(p, grads, state, lr)
    return loss, gnorm, raw


def ddp_train_step(p, tb, state, shards, mask, lr, num_heads, depth):
    """SURVEY 8(e) data-parallel semantics (no reference counterpart): each rank runs the
    reference forward/backward on ITS batch (attention local to that batch), gradients
    are averaged over ranks, then ONE clip + AdamW."""
    W = len(shards)
    acc, losses = None, []
    for (x, x_cond, noise, ts) in shards:
        loss, g, _ = loss_and_grads(p, tb, x, x_cond, noise, ts, mask, num_heads, depth)
        losses.append(loss)
        acc = g if acc is None else {k: acc[k] + g[k] for k in g}
    grads = {k: v / W for k, v in acc.items()}
    raw = {k: g.clone() for k, g in grads.items()}
    gnorm = clip_grads_(grads, 1.0)
    adamw_step_(p, grads, state, lr)
    return torch.stack(losses).mean(), gnorm, raw


# --------------------------------------------------------------------------
# Deterministic weights / synthetic data shared by goldens, tests and bench
# --------------------------------------------------------------------------

def param_shapes(G, H, depth, classes=6):
    """State-dict keys/shapes of DiT_stDiff in registration order (stDiff_model.py:176-196)."""
    s = [("in_layer.0.weight", (H, G)), ("in_layer.0.bias", (H,)),
         ("cond_layer.0.weight", (H, G)), ("cond_layer.0.bias", (H,)),
         ("condi_emb.weight", (classes, H)),
         ("time_emb.mlp.0.weight", (H, 256)), ("time_emb.mlp.0.bias", (H,)),
         ("time_emb.mlp.2.weight", (H, H)), ("time_emb.mlp.2.bias", (H,))]
    for k in range(depth):
        b = f"blks.{k}."
        s += [(b + "attn.qkv.weight", (3 * H, H)), (b + "attn.qkv.bias", (3 * H,)),
              (b + "attn.proj.weight", (H, H)), (b + "attn.proj.bias", (H,)),
              (b + "mlp.fc1.weight", (4 * H, H)), (b + "mlp.fc1.bias", (4 * H,)),
              (b + "mlp.fc2.weight", (H, 4 * H)), (b + "mlp.fc2.bias", (H,)),
              (b + "adaLN_modulation.1.weight", (6 * H, H)), (b + "adaLN_modulation.1.bias", (6 * H,))]
    s += [("out_layer.linear.weight", (G, H)), ("out_layer.linear.bias", (G,)),
          ("out_layer.adaLN_modulation.1.weight", (2 * H, H)), ("out_layer.adaLN_modulation.1.bias", (2 * H,))]
    return s


def make_params(G, H, depth, seed, classes=6):
    """Re-randomised weights (a fresh reference model outputs exactly 0 because its adaLN
    and output layers are zero-initialised, stDiff_model.py:218-229).  Weight matrices
    ~N(0, 1/fan_in) (x0.5 for adaLN so the modulation stays O(0.5)), biases ~N(0, 0.02)."""
    rng = np.random.default_rng(seed)
    p = {}
    for name, shp in param_shapes(G, H, depth, classes):
        a = rng.standard_normal(shp, dtype=np.float32)
        if name.endswith(".bias"):
            a *= 0.02
        elif name == "condi_emb.weight":
            a *= 0.02
        else:
            a *= (0.5 if "adaLN" in name else 1.0) / math.sqrt(shp[1])
        p[name] = torch.from_numpy(a)
    return p


def synth_expression(n, g, seed):
    """SURVEY 8(d) synthetic matrix: Poisson-LogNormal counts -> normalize_total(1e4) ->
    log1p -> per-row max-abs scale; fp32 in [0, 1] (test-stDiff.py:47-55)."""
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
    """mask of test-stDiff.py:73,92-96 without sklearn: KFold(shuffle=True, random_state)
    permutes arange(g) with RandomState(seed) and cuts it into n_splits consecutive folds."""
    idx = np.arange(g)
    np.random.RandomState(seed).shuffle(idx)
    sizes = np.full(n_splits, g // n_splits, dtype=int)
    sizes[: g % n_splits] += 1
    start = int(sizes[:fold].sum())
    test = idx[start:start + sizes[fold]]
    m = np.ones((g,), dtype=np.float32)
    m[test] = 0
    return m
