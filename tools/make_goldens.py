#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REAL reference (build container only).

Imports /root/reference/model/* (never copied, never shipped) with two import stubs for
packages that are not installed here and cannot be fetched: timm's ``Mlp`` (semantics of
timm 0.6.12: fc1 -> act -> drop -> fc2 -> drop) and ``ray.air.session`` (only touched when
is_tune=True).  Every vector is produced by the reference code itself; the oracle
(oracle/cpu_ref.py) is run beside it and must agree, otherwise this script fails.

Usage:  python tools/make_goldens.py            (writes tests/golden/)
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

from oracle import cpu_ref as O  # noqa: E402


def _install_stubs():
    class Mlp(nn.Module):
        def __init__(self, in_features, hidden_features=None, out_features=None,
                     act_layer=nn.GELU, drop=0.0):
            super().__init__()
            out_features = out_features or in_features
            hidden_features = hidden_features or in_features
            self.fc1 = nn.Linear(in_features, hidden_features)
            self.act = act_layer()
            self.drop1 = nn.Dropout(drop)
            self.fc2 = nn.Linear(hidden_features, out_features)
            self.drop2 = nn.Dropout(drop)

        def forward(self, x):
            return self.drop2(self.fc2(self.drop1(self.act(self.fc1(x)))))

    timm = types.ModuleType("timm")
    models = types.ModuleType("timm.models")
    vt = types.ModuleType("timm.models.vision_transformer")
    vt.Mlp = Mlp
    vt.PatchEmbed = object
    vt.Attention = object
    timm.models = models
    models.vision_transformer = vt
    sys.modules.update({"timm": timm, "timm.models": models, "timm.models.vision_transformer": vt})
    ray = types.ModuleType("ray")
    air = types.ModuleType("ray.air")
    session = types.ModuleType("ray.air.session")
    session.report = lambda *a, **k: None
    air.session = session
    ray.air = air
    sys.modules.update({"ray": ray, "ray.air": air, "ray.air.session": session})


_install_stubs()
sys.path.insert(0, REF)
# the repository's own top-level `model/` package (the import shim over stdiff_amd) would win over the reference's
# namespace package of the same name: bind the name `model` to the reference's directory explicitly
_refpkg = types.ModuleType("model")
_refpkg.__path__ = [os.path.join(REF, "model")]
sys.modules["model"] = _refpkg
from model.stDiff_model import DiT_stDiff  # noqa: E402
from model.stDiff_scheduler import NoiseScheduler  # noqa: E402
from model import stDiff_train as ref_train  # noqa: E402
from model import sample as ref_sample  # noqa: E402
from model import stDiff_scheduler as ref_sched  # noqa: E402

CPU = torch.device("cpu")


def ref_model(G, H, depth, heads, seed):
    m = DiT_stDiff(input_size=G, hidden_size=H, depth=depth, dit_type="dit", num_heads=heads,
                   classes=6, mlp_ratio=4.0)
    p = O.make_params(G, H, depth, seed)
    assert [k for k, _ in m.state_dict().items()] == list(p.keys()), "state-dict key order drifted"
    m.load_state_dict(p, strict=True)
    return m.eval(), p


def maxdiff(a, b):
    return float((a - b).abs().max())


def check(name, a, b, tol):
    d = maxdiff(torch.as_tensor(a), torch.as_tensor(b))
    s = float(torch.as_tensor(b).abs().max())
    print(f"  oracle-vs-reference {name}: max|d|={d:.3e} (scale {s:.3e})")
    assert d <= tol * max(s, 1.0), (name, d, s)


def wsum(p):
    return np.array([float(sum(v.double().abs().sum() for v in p.values()))])


def data(n, g, seed):
    x = torch.from_numpy(O.synth_expression(n, g, seed)) * 2 - 1
    m = O.kfold_mask(g, 0)
    xc = torch.from_numpy((O.synth_expression(n, g, seed) * m)) * 2 - 1
    return x, xc, m


# ---------------------------------------------------------------- G1 / G2
FWD_CASES = {
    # name: (N, G, H, depth, heads, seed)
    "A": (96, 40, 128, 2, 4, 11),      # d = 32
    "B": (80, 48, 512, 1, 16, 12),     # d = 32, H = 512 (config 1 family)
    "C": (70, 100, 1024, 1, 16, 13),   # d = 64 (config 2/4/5 family)
    "D": (48, 64, 2048, 1, 16, 14),    # d = 128 (config 3 family)
}


def g1_g2():
    for name, (N, G, H, depth, heads, seed) in FWD_CASES.items():
        print(f"G1 forward case {name}")
        m, p = ref_model(G, H, depth, heads, seed)
        x, xc, _ = data(N, G, 100 + seed)
        rng = np.random.default_rng(seed)
        x = x + torch.from_numpy(rng.standard_normal((N, G), dtype=np.float32)) * 0.3
        t = torch.from_numpy(rng.integers(0, 1000, size=(N,))).long()
        taps = {}
        hooks = [m.blks[0].register_forward_hook(lambda mod, i, o: taps.__setitem__("h_blk0", o.detach()))]
        with torch.no_grad():
            eps = m(x, t, xc)
            c = m.time_emb(t) + m.cond_layer(xc)
        for h in hooks:
            h.remove()
        ot = {}
        with torch.no_grad():
            eo = O.dit_forward(p, x, t, xc, heads, depth, taps=ot)
        check("eps", eo, eps, 2e-6)
        check("c", ot["c"], c, 2e-6)
        check("h_blk0", ot["h_blk0"], taps["h_blk0"], 2e-6)
        out = dict(x=x.numpy(), t=t.numpy(), y=xc.numpy(), eps=eps.numpy(), c=c.numpy(),
                   h_blk0=taps["h_blk0"].numpy(), cfg=np.array([N, G, H, depth, heads, seed]),
                   wsum=wsum(p))
        if name == "A":
            # G2: same rows, same t, chunked 40+40+16 (batch-coupled attention, partial chunk)
            tt = torch.full((N,), 437, dtype=torch.long)
            with torch.no_grad():
                one = m(x, tt, xc)
                parts = torch.cat([m(x[a:b], tt[a:b], xc[a:b]) for a, b in O.chunk_bounds(N, 40)])
                oparts = O.denoise_chunked(p, x, 437, xc, 40, heads, depth)
            assert maxdiff(one, parts) > 1e-3, "chunking must change the result"
            check("chunked", oparts, parts, 2e-6)
            out.update(g2_t=np.array([437]), g2_one=one.numpy(), g2_chunks=parts.numpy(), g2_batch=np.array([40]))
        np.savez(os.path.join(OUT, f"g1_forward_{name}.npz"), **out)


# ---------------------------------------------------------------- G3 / G4
def g3_g4():
    print("G3 scheduler tables")
    out = {}
    for sched, T in (("cosine", 20), ("cosine", 1000), ("cosine", 1500), ("linear", 1000), ("quadratic", 1000)):
        ns = NoiseScheduler(num_timesteps=T, beta_schedule=sched)
        tb = O.make_tables(T, schedule=sched)
        for n in O.TABLE_NAMES:
            ref = getattr(ns, n)
            assert ref.dtype == torch.float32 and ref.device.type == "cpu"
            assert torch.equal(ref, tb[n]), (sched, T, n)
            out[f"{sched}_{T}_{n}"] = ref.numpy()
        var = np.array([float(ns.get_variance(torch.tensor(t))) for t in range(T)], dtype=np.float64)
        ov = np.array([float(O.variance(tb, t)) for t in range(T)], dtype=np.float64)
        assert np.array_equal(var, ov)
        out[f"{sched}_{T}_variance"] = var
        assert len(ns) == T
    np.savez(os.path.join(OUT, "g3_tables.npz"), **out)

    print("G4 scheduler.step with injected z")
    T = 1000
    ns = NoiseScheduler(num_timesteps=T, beta_schedule="cosine")
    tb = O.make_tables(T, schedule="cosine")
    rng = np.random.default_rng(4)
    N, G = 64, 50
    out = {}
    real_randn_like = torch.randn_like
    for t in (T - 1, T // 2, 1, 0):
        scale = {T - 1: 1.0, T // 2: 0.8, 1: 0.5, 0: 0.5}[t]
        x_t = torch.from_numpy(rng.standard_normal((N, G), dtype=np.float32)) * scale
        eps = torch.from_numpy(rng.standard_normal((N, G), dtype=np.float32))
        z = torch.from_numpy(rng.standard_normal((N, G), dtype=np.float32))
        ref_sched.torch.randn_like = lambda a, _z=z: _z
        try:
            prev, x0 = ns.step(eps, torch.tensor(t).long(), x_t, model_pred_type="noise")
            prev_xs, x0_xs = ns.step(eps, torch.tensor(t).long(), x_t, model_pred_type="x_start")
        finally:
            ref_sched.torch.randn_like = real_randn_like
        op, ox0 = O.step(tb, eps, t, x_t, z, "noise")
        assert torch.equal(op, prev) and torch.equal(ox0, x0), t
        opx, _ = O.step(tb, eps, t, x_t, z, "x_start")
        assert torch.equal(opx, prev_xs)
        frac = float(((x0 == 1) | (x0 == -1)).float().mean())
        print(f"  t={t}: clamped fraction {frac:.3f}")
        out.update({f"t{t}_x": x_t.numpy(), f"t{t}_eps": eps.numpy(), f"t{t}_z": z.numpy(),
                    f"t{t}_prev": prev.numpy(), f"t{t}_x0": x0.numpy(), f"t{t}_prev_xstart": prev_xs.numpy()})
    # add_noise / undo
    x0 = torch.from_numpy(rng.standard_normal((N, G), dtype=np.float32)).clamp(-1, 1)
    nz = torch.from_numpy(rng.standard_normal((N, G), dtype=np.float32))
    ts = torch.from_numpy(rng.integers(1, T, size=(N,))).long()
    an = ns.add_noise(x0, nz, ts)
    assert torch.equal(an, O.add_noise(tb, x0, nz, ts))
    out.update(an_x0=x0.numpy(), an_noise=nz.numpy(), an_t=ts.numpy(), an_out=an.numpy())
    try:
        ns.step(eps, torch.tensor(3), x_t, model_pred_type="v")
        raise AssertionError("expected NotImplementedError")
    except NotImplementedError:
        pass
    np.savez(os.path.join(OUT, "g4_step.npz"), **out)


# ---------------------------------------------------------------- G5
def g5():
    print("G5 full sample_stDiff, T=20, N=96, batch 40, injected x_T and z_t")
    N, G, H, depth, heads, seed = 96, 40, 128, 2, 4, 11
    T, B = 20, 40
    m, p = ref_model(G, H, depth, heads, seed)
    _, _, mask = data(N, G, 111)
    st = O.synth_expression(N, G, 21)
    gt = (st * mask) * 2 - 1
    rng = np.random.default_rng(5)
    x_T = torch.from_numpy(rng.standard_normal((N, G), dtype=np.float32))
    zs = {t: torch.from_numpy(rng.standard_normal((N, G), dtype=np.float32)) for t in range(T - 1, 0, -1)}
    ns = NoiseScheduler(num_timesteps=T, beta_schedule="cosine")
    loader = torch.utils.data.DataLoader(
        torch.utils.data.TensorDataset(torch.from_numpy(gt), torch.from_numpy(gt)), batch_size=B, shuffle=False)

    rec = {"eps": {}, "x_out": {}}
    zq = [zs[t] for t in range(T - 1, 0, -1)]
    real_randn, real_randn_like = torch.randn, torch.randn_like
    ref_sample.torch.randn = lambda *a, **k: x_T.clone()
    ref_sched.torch.randn_like = lambda a: zq.pop(0)
    # record per-step eps through the reference's own chunked denoiser call
    orig_mss = ref_sample.model_sample_stDiff

    def spy(model, device, dataloader, total_sample, time, is_condi, condi_flag):
        e = orig_mss(model, device, dataloader, total_sample, time, is_condi, condi_flag)
        rec["eps"][int(time)] = e.clone()
        rec.setdefault("x_in", {})[int(time)] = total_sample.clone()
        return e
    ref_sample.model_sample_stDiff = spy
    try:
        final = ref_sample.sample_stDiff(m, loader, ns, mask=mask, gt=gt, device=CPU, num_step=T,
                                         sample_shape=(N, G), is_condi=True, sample_intermediate=T,
                                         model_pred_type="noise", is_classifier_guidance=False, omega=0.2)
    finally:
        torch.randn, torch.randn_like = real_randn, real_randn_like
        ref_sample.model_sample_stDiff = orig_mss
    assert not zq
    assert isinstance(final, np.ndarray) and final.dtype == np.float32 and final.shape == (N, G)
    orec = {}
    tb = O.make_tables(T, schedule="cosine")
    of = O.sample(p, tb, gt, mask, B, heads, depth, x_T, lambda t: zs[t], T, T, "noise", record=orec)
    # teacher-forced oracle eps must match the reference step by step
    for t in range(T - 1, -1, -1):
        with torch.no_grad():
            e = O.denoise_chunked(p, rec["x_in"][t], t, torch.from_numpy(gt), B, heads, depth)
        check(f"eps[t={t}] (teacher forced)", e, rec["eps"][t], 3e-6)
    check("final (free running)", of, torch.from_numpy(final), 1e-4)
    np.savez(os.path.join(OUT, "g5_sample.npz"),
             cfg=np.array([N, G, H, depth, heads, seed, T, B]), mask=mask, gt=gt, x_T=x_T.numpy(),
             z=np.stack([zs[t].numpy() for t in range(T - 1, 0, -1)]),
             eps=np.stack([rec["eps"][t].numpy() for t in range(T - 1, -1, -1)]),
             x_in=np.stack([rec["x_in"][t].numpy() for t in range(T - 1, -1, -1)]),
             final=final, wsum=wsum(p))
    # early termination: sample_intermediate=5 keeps only the first 5 timesteps (sample.py:63-64)
    zq = [zs[t] for t in range(T - 1, T - 6, -1)]
    ref_sample.torch.randn = lambda *a, **k: x_T.clone()
    ref_sched.torch.randn_like = lambda a: zq.pop(0)
    try:
        part = ref_sample.sample_stDiff(m, loader, ns, mask=mask, gt=gt, device=CPU, num_step=T,
                                        sample_shape=(N, G), is_condi=True, sample_intermediate=5)
    finally:
        torch.randn, torch.randn_like = real_randn, real_randn_like
    op = O.sample(p, tb, gt, mask, B, heads, depth, x_T, lambda t: zs[t], T, 5)
    check("partial (sample_intermediate=5)", op, torch.from_numpy(part), 1e-4)
    np.savez(os.path.join(OUT, "g5_partial.npz"), final=part, steps=np.array([5]))


# ---------------------------------------------------------------- G6 / G7
def _patched_train(m, batches, mask, lr, T, draws):
    """Run the reference normal_train_stDiff for one epoch over ``batches`` with the RNG
    draws replaced by the recorded tensors in ``draws`` [(noise, timesteps), ...]."""
    q = list(draws)
    cur = {}
    real_randn, real_randint = torch.randn, torch.randint

    def fake_randn(*a, **k):
        cur["pair"] = q.pop(0)
        return cur["pair"][0].clone()

    def fake_randint(*a, **k):
        return cur["pair"][1].clone()
    ref_train.torch.randn = fake_randn
    ref_train.torch.randint = fake_randint
    try:
        ref_train.normal_train_stDiff(m, batches, lr=lr, num_epoch=1, pred_type="noise", diffusion_step=T,
                                      device=CPU, is_tqdm=False, is_tune=False, mask=mask)
    finally:
        torch.randn, torch.randint = real_randn, real_randint


def g6_g7():
    print("G6 three consecutive train steps (injected eps / timesteps)")
    C, G, H, depth, heads, seed = 64, 40, 128, 2, 4, 16
    T, lr = 1000, 0.00016046744893538737
    m, p = ref_model(G, H, depth, heads, seed)
    m.train()
    x, xc, mask = data(3 * C, G, 31)
    rng = np.random.default_rng(6)
    draws = [(torch.from_numpy(rng.standard_normal((C, G), dtype=np.float32)),
              torch.from_numpy(rng.integers(1, T, size=(C,))).long()) for _ in range(3)]
    batches = [(x[i * C:(i + 1) * C], xc[i * C:(i + 1) * C]) for i in range(3)]

    # reference: capture losses, pre-clip grad norm and raw grads by hooking clip_grad_norm_
    cap = {"gn": [], "grads": [], "loss": []}
    real_clip = nn.utils.clip_grad_norm_

    def spy_clip(params, max_norm, *a, **k):
        params = list(params)
        cap["grads"].append({n: q.grad.detach().clone() for n, q in m.named_parameters() if q.grad is not None})
        assert [q.grad is None for n, q in m.named_parameters() if n == "condi_emb.weight"] == [True]
        gn = real_clip(params, max_norm, *a, **k)
        cap["gn"].append(float(gn))
        return gn
    real_mse = nn.MSELoss.forward

    def spy_mse(self, a, b):
        l = real_mse(self, a, b)
        cap["loss"].append(float(l))
        return l
    ref_train.nn.utils.clip_grad_norm_ = spy_clip
    nn.MSELoss.forward = spy_mse
    try:
        _patched_train(m, batches, mask, lr, T, draws)
    finally:
        nn.utils.clip_grad_norm_ = real_clip
        nn.MSELoss.forward = real_mse
    ref_post = {k: v.detach().clone() for k, v in m.state_dict().items()}

    tb = O.make_tables(T, schedule="cosine")
    po = {k: v.clone() for k, v in p.items()}
    state = {}
    out = dict(cfg=np.array([C, G, H, depth, heads, seed, T]), lr=np.array([lr]), mask=mask,
               x=x.numpy(), x_cond=xc.numpy(), noise=np.stack([d[0].numpy() for d in draws]),
               timesteps=np.stack([d[1].numpy() for d in draws]),
               loss=np.array(cap["loss"]), gnorm=np.array(cap["gn"]), wsum=wsum(p))
    for i in range(3):
        loss, gn, raw = O.train_step(po, tb, state, batches[i][0], batches[i][1], draws[i][0], draws[i][1],
                                     mask, lr, heads, depth)
        print(f"  step {i}: loss ref {cap['loss'][i]:.6f} oracle {float(loss):.6f}; "
              f"gnorm ref {cap['gn'][i]:.5f} oracle {float(gn):.5f}")
        assert abs(float(loss) - cap["loss"][i]) <= 2e-6 * max(1, cap["loss"][i])
        assert abs(float(gn) - cap["gn"][i]) <= 1e-4 * cap["gn"][i]
        for n, g in cap["grads"][i].items():
            d = maxdiff(raw[n], g)
            assert d <= 2e-5 * max(float(g.abs().max()), 1e-3), (i, n, d)
    for n in O.trainable_names(p):
        # k-bias rows of qkv.bias have a mathematically-zero gradient (softmax is invariant to a
        # per-query constant), so Adam normalises pure rounding noise there: allow a few 1e-6.
        check(f"post-step weight {n}", po[n], ref_post[n], 5e-6 if n.endswith("qkv.bias") else 2e-6)
    assert torch.equal(ref_post["condi_emb.weight"], p["condi_emb.weight"])
    # fixtures: every grad's L2 norm for all 3 steps, full grads of step 0 for a subset, full post weights subset
    names = O.trainable_names(p)
    out["grad_names"] = np.array(names)
    out["grad_l2"] = np.array([[float(cap["grads"][i][n].norm()) for n in names] for i in range(3)])
    out["post_l2"] = np.array([float(ref_post[n].double().norm()) for n in names])
    out["delta_l2"] = np.array([float((ref_post[n] - p[n]).double().norm()) for n in names])
    full = ["in_layer.0.weight", "cond_layer.0.bias", "time_emb.mlp.0.weight", "blks.0.attn.qkv.weight",
            "blks.0.attn.qkv.bias", "blks.1.attn.proj.weight", "blks.1.mlp.fc1.bias", "blks.0.mlp.fc2.weight",
            "blks.1.adaLN_modulation.1.weight", "blks.0.adaLN_modulation.1.bias",
            "out_layer.linear.weight", "out_layer.linear.bias", "out_layer.adaLN_modulation.1.weight"]
    for n in full:
        out["g0/" + n] = cap["grads"][0][n].numpy()
        out["post/" + n] = ref_post[n].numpy()
    np.savez(os.path.join(OUT, "g6_train.npz"), **out)

    print("G7 data-parallel semantics, W=2 (reference grads of 2 batches averaged, one optimizer step)")
    m2, p2 = ref_model(G, H, depth, heads, seed)
    m2.train()
    ns = NoiseScheduler(num_timesteps=T, beta_schedule="cosine")
    mk = torch.tensor(mask)
    acc = None
    for i in range(2):
        m2.zero_grad()
        xb, cb = batches[i]
        nz, ts = draws[i]
        x_t = ns.add_noise(xb, nz, ts)
        x_noisy = x_t * (1 - mk) + xb * mk
        pred = m2(x_noisy, t=ts, y=cb)
        loss = nn.MSELoss()(nz * (1 - mk), pred * (1 - mk))
        loss.backward()
        g = {n: q.grad.detach().clone() for n, q in m2.named_parameters() if q.grad is not None}
        acc = g if acc is None else {n: acc[n] + g[n] for n in g}
    opt = torch.optim.AdamW(m2.parameters(), lr=lr, weight_decay=0)
    for n, q in m2.named_parameters():
        q.grad = None if n == "condi_emb.weight" else (acc[n] / 2).clone()
    gn = float(nn.utils.clip_grad_norm_(m2.parameters(), 1.0))
    opt.step()
    ref_post2 = {k: v.detach().clone() for k, v in m2.state_dict().items()}
    po = {k: v.clone() for k, v in p2.items()}
    _, ogn, _ = O.ddp_train_step(po, tb, {}, [(batches[i][0], batches[i][1], draws[i][0], draws[i][1]) for i in range(2)],
                                 mask, lr, heads, depth)
    assert abs(float(ogn) - gn) <= 1e-4 * gn
    for n in names:
        check(f"ddp post weight {n}", po[n], ref_post2[n], 5e-6 if n.endswith("qkv.bias") else 2e-6)
    o7 = dict(gnorm=np.array([gn]), post_l2=np.array([float(ref_post2[n].double().norm()) for n in names]),
              delta_l2=np.array([float((ref_post2[n] - p2[n]).double().norm()) for n in names]),
              grad_names=np.array(names))
    for n in full:
        o7["post/" + n] = ref_post2[n].numpy()
    np.savez(os.path.join(OUT, "g7_ddp.npz"), **o7)


# ---------------------------------------------------------------- G8: the 5-fold driver flow
def g8():
    """test-stDiff.py:62-187 (`diffusion_impute`) on a tiny problem.  The script itself cannot be imported (scanpy, CUDA
    default tensor type at module level), so its call sequence is followed here line by line with the REFERENCE's own
    DiT_stDiff / normal_train_stDiff / NoiseScheduler / sample_stDiff and sklearn's KFold; the DataLoader factory
    (process/data.py:64-75, needs anndata and a CUDA generator) is the same two torch calls on a CPU generator.
    Training runs for real (so the zero-initialised layers move), the trained state dicts are recorded -- the build's
    driver then LOADS them as checkpoints `stDiff%d.pt`, which is the reference's own resume path (:138,151) -- and the
    sampling draws are recorded so that the build can inject them."""
    from sklearn.model_selection import KFold
    from torch.utils.data import DataLoader, TensorDataset
    print("G8 diffusion_impute flow: 5 folds, reference train + sample")
    n_sc, n_st, G, H, depth, heads, B, T = 64, 48, 20, 64, 1, 2, 32, 12
    lr, num_epoch, n_splits, rand = 0.00016046744893538737, 3, 5, 0
    data_seq_array = O.synth_expression(n_sc, G, 41)
    data_spatial_array = O.synth_expression(n_st, G, 42)

    def get_data_loader(data_ary, cell_type, batch_size=512, is_shuffle=True):
        ds = TensorDataset(torch.from_numpy(data_ary.astype(np.float32)), torch.from_numpy(cell_type.astype(np.float32)))
        gen = torch.Generator(device="cpu")
        return DataLoader(ds, batch_size=batch_size, shuffle=is_shuffle, drop_last=False, generator=gen)

    kf = KFold(n_splits=n_splits, shuffle=True, random_state=rand)                      # :73
    torch.manual_seed(10)                                                               # :75
    idx = 1
    all_pred_res = np.zeros_like(data_spatial_array)
    out = dict(cfg=np.array([n_sc, n_st, G, H, depth, heads, B, T, num_epoch, n_splits, rand]), lr=np.array([lr]),
               seq=data_seq_array, st=data_spatial_array)
    real_randn, real_randn_like = torch.randn, torch.randn_like
    for train_ind, test_ind in kf.split(np.arange(G)):                                  # :79
        cell_num, gene_num = data_spatial_array.shape
        mask = np.ones((gene_num,), dtype="float32")                                    # :92-96
        mask[test_ind] = 0
        seq, st = data_seq_array, data_spatial_array
        data_seq_masked, data_spatial_masked = seq * mask, st * mask                    # :100-101
        seq = seq * 2 - 1                                                               # :103-108
        data_seq_masked = data_seq_masked * 2 - 1
        data_spatial_masked = data_spatial_masked * 2 - 1
        dataloader = get_data_loader(seq, data_seq_masked, batch_size=B, is_shuffle=True)
        torch.manual_seed(1202)                                                         # :116-119
        model = DiT_stDiff(input_size=gene_num, hidden_size=H, depth=depth, num_heads=heads, classes=6,
                           mlp_ratio=4.0, dit_type="dit")                               # :121-129
        model.to(CPU)
        model.train()
        ref_train.normal_train_stDiff(model, dataloader=dataloader, lr=lr, num_epoch=num_epoch, diffusion_step=T,
                                      device=CPU, pred_type="noise", mask=mask, is_tqdm=False)   # :140-147
        sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
        assert float(sd["out_layer.linear.weight"].abs().max()) > 0                     # training moved the zero layers
        gt = data_spatial_masked                                                        # :153
        noise_scheduler = NoiseScheduler(num_timesteps=T, beta_schedule="cosine")
        dataloader = get_data_loader(data_spatial_masked, data_spatial_masked, batch_size=B, is_shuffle=False)
        model.eval()
        rng = np.random.default_rng(800 + idx)
        x_T = torch.from_numpy(rng.standard_normal((cell_num, gene_num), dtype=np.float32))
        z = [torch.from_numpy(rng.standard_normal((cell_num, gene_num), dtype=np.float32)) for _ in range(T - 1)]
        zq = list(z)
        ref_sample.torch.randn = lambda *a, **k: x_T.clone()
        ref_sched.torch.randn_like = lambda a: zq.pop(0)
        try:
            imputation = ref_sample.sample_stDiff(model, device=CPU, dataloader=dataloader,
                                                  noise_scheduler=noise_scheduler, mask=mask, gt=gt, num_step=T,
                                                  sample_shape=(cell_num, gene_num), is_condi=True,
                                                  sample_intermediate=T, model_pred_type="noise",
                                                  is_classifier_guidance=False, omega=0.2)     # :167-179
        finally:
            torch.randn, torch.randn_like = real_randn, real_randn_like
        assert not zq
        all_pred_res[:, test_ind] = imputation[:, test_ind]                             # :181
        # the oracle reproduces the reference's sampling of this fold
        zs = {t: z[i] for i, t in enumerate(range(T - 1, 0, -1))}
        of = O.sample({k: v for k, v in sd.items()}, O.make_tables(T, schedule="cosine"), gt.astype(np.float32), mask, B,
                      heads, depth, x_T, lambda t: zs[t], T, T)
        check(f"G8 fold {idx} sample (oracle vs reference)", of, torch.from_numpy(imputation), 1e-4)
        for k, v in sd.items():
            out[f"sd{idx}/{k}"] = v.numpy()
        out[f"x_T{idx}"] = x_T.numpy()
        out[f"z{idx}"] = np.stack([t.numpy() for t in z])
        out[f"test_ind{idx}"] = test_ind
        idx += 1
    impu = (all_pred_res + 1) / 2                                                       # :184
    out["impu"] = impu.astype(np.float32)
    np.savez(os.path.join(OUT, "g8_driver.npz"), **out)


# ---------------------------------------------------------------- G9: the evaluation block's per-gene metrics
def _reference_metric_code():
    """cal_ssim, scale_max, scale_z_score, scale_plus and CalculateMeteics of test-stDiff.py:203-364, EXECUTED from the
    reference's own file: the script cannot be imported (scanpy, argparse and a full experiment at module level), so the
    definitions are taken out of its syntax tree by name and compiled as they stand -- nothing is retyped here."""
    import ast
    import pandas as pd
    import scipy.stats as st
    path = os.path.join(REF, "test-stDiff.py")
    tree = ast.parse(open(path).read(), filename=path)
    want = {"cal_ssim", "scale_max", "scale_z_score", "scale_plus", "CalculateMeteics"}
    body = [n for n in tree.body if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name in want]
    assert {n.name for n in body} == want
    ns = {"np": np, "pd": pd, "st": st}
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), ns)
    return ns


def g9():
    import tempfile
    import warnings
    import pandas as pd
    print("G9 per-gene metrics: the reference's CalculateMeteics.SSIM / SPCC / JS / RMSE (test-stDiff.py:254-364)")
    ns = _reference_metric_code()
    rng = np.random.default_rng(9)
    N, G = 300, 14
    lam = rng.lognormal(-0.5, 1.0, size=G)
    raw = rng.poisson(lam[None, :] * rng.lognormal(0.0, 0.5, size=(N, 1))).astype(np.float32)   # tie-heavy counts
    # imputed: a noisy monotone function of the counts, on a 1/1024 grid (exact in fp32 AND as the decimal text of the
    # CSV the reference reads the imputed matrix back from), mostly positive with exact zeros
    imp = np.log1p(raw) * 0.7 + rng.normal(0, 0.25, size=(N, G))
    imp = (np.round(np.clip(imp, 0, None) * 1024) / 1024).astype(np.float32)
    # degenerate columns, as real panels have them (a gene nobody expresses; a saturated one)
    raw[:, 9] = 0.0                      # measured gene all zero
    raw[:, 10] = 3.0                     # measured gene constant
    imp[:, 11] = 0.5                     # imputed gene constant
    raw[:, 12] = 0.0
    imp[:, 12] = 0.0                     # both all zero
    imp[:, 13] = 0.0                     # imputed all zero, measured regular
    genes = [f"Gene{j}" for j in range(G)]
    with tempfile.TemporaryDirectory() as td:
        csv = os.path.join(td, "diffusion_impute.csv")
        pd.DataFrame(imp.astype(np.float64), columns=genes).to_csv(csv, header=1, index=1)        # :198-199
        back = pd.read_csv(csv, header=0, index_col=0).to_numpy()
        assert np.array_equal(back.astype(np.float32), imp) and np.array_equal(back, imp.astype(np.float64))
        cm = ns["CalculateMeteics"](raw, genes, impute_count_file=csv, prefix=os.path.join(td, "x"), metric=None)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            with np.errstate(all="ignore"):
                res = {"SSIM": cm.SSIM(cm.raw_count, cm.impute_count), "SPCC": cm.SPCC(cm.raw_count, cm.impute_count),
                       "JS": cm.JS(cm.raw_count, cm.impute_count), "RMSE": cm.RMSE(cm.raw_count, cm.impute_count)}
    out = dict(raw=raw, imputed=imp)
    for k, df in res.items():
        assert list(df.columns) == [g.upper() for g in genes]
        out[k] = df.to_numpy(dtype=np.float64).reshape(G)
        print(f"  {k}: " + " ".join(f"{v:.4g}" for v in out[k]))
    np.savez(os.path.join(OUT, "g9_metrics.npz"), **out)


# ---------------------------------------------------------------- G10: scale() of process/data.py
def g10():
    """process/data.py:79-83 `scale` imported from the reference (stubs for the absent anndata / scanpy packages only;
    sklearn's MaxAbsScaler, which does the arithmetic, is the real one) and run on an object with `.X`.
    normalize_total / log1p are scanpy's and stay unpinned."""
    import importlib.util
    print("G10 scale(): MaxAbsScaler().fit_transform(X.T).T of the reference's process/data.py")
    for name in ("anndata", "scanpy"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.AnnData = object
            sys.modules[name] = m
    import scipy.sparse
    if not hasattr(scipy.sparse, "csr"):
        scipy.sparse.csr = types.ModuleType("scipy.sparse.csr")
    spec = importlib.util.spec_from_file_location("ref_process_data", os.path.join(REF, "process", "data.py"))
    ref_data = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_data)
    rng = np.random.default_rng(10)
    N, G = 257, 75
    counts = rng.poisson(rng.lognormal(0, 1, size=(1, G)) * rng.lognormal(0, 0.5, size=(N, 1))).astype(np.float32)
    X = np.log1p(counts / np.maximum(counts.sum(1, keepdims=True), 1) * 1e4).astype(np.float32)
    X[5] = 0.0                                       # a cell without counts: divides by 1
    X[6] *= -1.0                                     # max ABS value
    X[7, 3] = -2.0 * X[7].max() - 1.0
    obj = types.SimpleNamespace(X=X.copy())
    got = ref_data.scale(obj).X
    assert got.dtype == np.float32 and got.shape == X.shape
    np.savez(os.path.join(OUT, "g10_scale.npz"), X=X, scaled=got)
    print(f"  scaled: row max-abs in [{np.abs(got).max(1).min():.3f}, {np.abs(got).max(1).max():.3f}]")


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    only = sys.argv[1] if len(sys.argv) > 1 else None        # e.g. `python tools/make_goldens.py g8`: one group only
    if only in (None, "g3", "g4"):
        g3_g4()
    if only in (None, "g1", "g2"):
        g1_g2()
    if only in (None, "g5"):
        g5()
    if only in (None, "g6", "g7"):
        g6_g7()
    if only in (None, "g8"):
        g8()
    if only in (None, "g9"):
        g9()
    if only in (None, "g10"):
        g10()
    print("goldens written to", OUT)
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(f"  {f}: {os.path.getsize(os.path.join(OUT, f)) / 1024:.0f} KiB")
