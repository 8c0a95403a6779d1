"""GPU: training path (forward with saved activations, hand-written backward, clip + AdamW) against the golden
vectors recorded from the reference (G6, G7) and against the oracle's autograd on fresh inputs."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import cpu_ref as O  # noqa: E402

DEV = "cuda:0"


def relerr(a, b):
    a, b = torch.as_tensor(a).float().cpu(), torch.as_tensor(b).float().cpu()
    return float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30)


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def build(G, H, depth, heads, seed, dtype):
    from stdiff_amd.model.stDiff_model import DiT_stDiff
    m = DiT_stDiff(G, H, depth, 'dit', heads, 6, compute_dtype=dtype)
    p = O.make_params(G, H, depth, seed)
    m.load_state_dict(p, strict=True)
    return m.to(DEV), p


# ---------------------------------------------------------------- backward kernels one by one
@pytest.mark.parametrize("dt", ["fp32", "bf16"])
def test_pointwise_backward_kernels(dt):
    from stdiff_amd import ops
    tdt = torch.bfloat16 if dt == "bf16" else torch.float32
    tol = 2e-2 if dt == "bf16" else 1e-5
    q = (lambda v: v.to(torch.bfloat16).float()) if dt == "bf16" else (lambda v: v)
    R, H = 70, 256
    # LN-modulate
    x, mod, dy = rnd(R, H, seed=1) * 1.5 + 0.2, rnd(R, 6 * H, seed=2, scale=0.5), rnd(R, H, seed=3)
    xr = x.clone().requires_grad_(True)
    sc = q(mod)[:, H:2 * H].clone().requires_grad_(True)
    sh = q(mod)[:, 0:H].clone().requires_grad_(True)
    y = F.layer_norm(xr, (H,), eps=1e-6) * (1 + sc) + sh
    y.backward(q(dy))
    md = mod.to(DEV).to(tdt)
    mean, rstd = torch.empty(R, device=DEV), torch.empty(R, device=DEV)
    ops.ln_modulate(x.to(DEV), md[:, 0:H], md[:, H:2 * H], mean=mean, rstd=rstd)
    dx = torch.full((R, H), 0.5, device=DEV)
    dmod = torch.zeros(R, 6 * H, device=DEV, dtype=tdt)
    ops.ln_modulate_bwd(dy.to(DEV).to(tdt), x.to(DEV), mean, rstd, md[:, H:2 * H], dx, dmod[:, 0:H], dmod[:, H:2 * H],
                        accumulate=True)
    assert relerr(dx - 0.5, xr.grad) < tol
    assert relerr(dmod[:, 0:H], sh.grad) < tol and relerr(dmod[:, H:2 * H], sc.grad) < tol
    # gate
    dh, br = rnd(R, H, seed=4), rnd(R, H, seed=5)
    dbr, dg = torch.empty(R, H, device=DEV, dtype=tdt), torch.empty(R, 6 * H, device=DEV, dtype=tdt)
    ops.gate_bwd(dh.to(DEV), md[:, 2 * H:3 * H], br.to(DEV).to(tdt), dbr, dg[:, 2 * H:3 * H])
    assert relerr(dbr, dh * q(mod)[:, 2 * H:3 * H]) < tol and relerr(dg[:, 2 * H:3 * H], dh * q(br)) < tol
    # gelu / silu
    u = rnd(R, 4 * H, seed=6).requires_grad_(True)
    dgl = rnd(R, 4 * H, seed=7)
    F.gelu(q(u.detach()).requires_grad_(True)).backward(q(dgl))
    ur = q(u.detach()).requires_grad_(True)
    F.gelu(ur).backward(q(dgl))
    du = torch.empty(R, 4 * H, device=DEV, dtype=tdt)
    ops.gelu_bwd(dgl.to(DEV).to(tdt), u.detach().to(DEV).to(tdt), du)
    assert relerr(du, ur.grad) < tol
    c = rnd(R, H, seed=8).requires_grad_(True)
    F.silu(c).backward(dh)
    dc = torch.full((R, H), 1.0, device=DEV)
    ops.silu_bwd(dh.to(DEV), c.detach().to(DEV), dc, accumulate=True)
    assert relerr(dc - 1.0, c.grad) < 1e-5
    # colsum
    db = torch.empty(H, device=DEV)
    ops.colsum(dy.to(DEV).to(tdt), db)
    assert relerr(db, q(dy).sum(0)) < 1e-5


@pytest.mark.parametrize("dt", ["fp32", "bf16"])
@pytest.mark.parametrize("d,heads,lens", [(32, 4, [96]), (64, 16, [512]), (64, 4, [200, 40, 1]), (128, 4, [130])])
def test_attention_bwd(dt, d, heads, lens):
    from stdiff_amd import ops
    tdt = torch.bfloat16 if dt == "bf16" else torch.float32
    q = (lambda v: v.to(torch.bfloat16).float()) if dt == "bf16" else (lambda v: v)
    H, R = d * heads, sum(lens)
    qkv = q(rnd(R, 3 * H, seed=d + R)).requires_grad_(True)
    dout = q(rnd(R, H, seed=1))
    outs, a = [], 0
    for n in lens:
        blk = qkv[a:a + n].reshape(n, 3, heads, d).permute(1, 2, 0, 3)
        s = (blk[0] @ blk[1].transpose(-2, -1)) * d ** -0.5
        outs.append((s.softmax(-1) @ blk[2]).transpose(0, 1).reshape(n, H))
        a += n
    ref = torch.cat(outs)
    ref.backward(dout)
    cu = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int32, device=DEV)
    qd = qkv.detach().to(DEV).to(tdt)
    lse = torch.empty(R, heads, device=DEV)
    out = ops.attention_fwd(qd, cu, max(lens), heads, lse=lse)
    dqkv = ops.attention_bwd(qd, out, dout.to(DEV).to(tdt), lse, cu, max(lens), heads)
    tol = 1e-5 if dt == "fp32" else 2e-2
    for i, nm in enumerate("qkv"):
        assert relerr(dqkv[:, i * H:(i + 1) * H], qkv.grad[:, i * H:(i + 1) * H]) < tol, nm


def test_clip_adamw_matches_torch():
    from stdiff_amd import ops
    n = 10007
    p0, g = rnd(n, seed=1), rnd(n, seed=2, scale=0.05)
    pt = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([pt], lr=1e-3, weight_decay=0.01)
    p = p0.to(DEV)
    m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    step, scal, ss = torch.zeros(1, dtype=torch.int32, device=DEV), torch.zeros(4, device=DEV), torch.zeros(1, device=DEV)
    for it in range(3):
        gi = g * (it + 1)
        pt.grad = gi.clone()
        gn = torch.nn.utils.clip_grad_norm_([pt], 1.0)
        opt.step()
        ss.zero_()
        gd = gi.to(DEV)
        ops.grad_sumsq(gd, ss)
        ops.clip_adamw(p, gd, m, v, ss, step, scal, 1e-3, weight_decay=0.01)
        assert abs(float(scal[3]) - float(gn)) < 1e-5 * float(gn)
        assert relerr(p, pt.detach()) < 2e-6
    assert int(step) == 3


# ---------------------------------------------------------------- whole train step vs the reference goldens
@pytest.mark.parametrize("use_graph,side,grouped", [(False, "1", "1"), (True, "1", "1"), (True, "0", "1"),
                                                    (False, "2", "1"), (True, "1", "0"), (False, "2", "0")])
def test_train_steps_golden_fp32(golden_dir, use_graph, side, grouped, monkeypatch):
    """3 consecutive reference iterations (G6); use_graph=True replays each as one captured hipGraph, whose weight
    gradients sit on parallel branches (side stream during capture; side="0": single chain; side="2" forks in the
    eager enqueue too, which exercises the fork / late-join ordering without a graph).  grouped="0": one job of three
    launches per linear layer instead of one stdiff_dw_group launch per block (the path of callers that do not pass
    the second set of dY buffers)."""
    from stdiff_amd.train_engine import TrainEngine
    monkeypatch.setenv("STDIFF_TRAIN_SIDE_STREAM", side)
    monkeypatch.setenv("STDIFF_TRAIN_GROUPED", grouped)
    g = np.load(os.path.join(golden_dir, "g6_train.npz"))
    C, G, H, depth, heads, seed, T = [int(v) for v in g["cfg"]]
    m, p0 = build(G, H, depth, heads, seed, "fp32")
    te = TrainEngine(m, lr=float(g["lr"][0]), diffusion_step=T, mask=g["mask"])
    names = [str(n) for n in g["grad_names"]]
    x, xc = torch.from_numpy(g["x"]).to(DEV), torch.from_numpy(g["x_cond"]).to(DEV)
    for i in range(3):
        loss, gn = te.step(x[i * C:(i + 1) * C].contiguous(), xc[i * C:(i + 1) * C].contiguous(),
                           noise=torch.from_numpy(g["noise"][i]).to(DEV),
                           timesteps=torch.from_numpy(g["timesteps"][i]).to(DEV), use_graph=use_graph)
        assert abs(float(loss) - g["loss"][i]) < 1e-5 * max(1.0, g["loss"][i]), i
        assert abs(float(gn) - g["gnorm"][i]) < 1e-4 * g["gnorm"][i], i
        l2 = np.array([float(te.g[n].norm()) for n in names])
        assert np.allclose(l2, g["grad_l2"][i], rtol=2e-3, atol=2e-6), (i, np.abs(l2 - g["grad_l2"][i]).max())
        if i == 0:
            for k in g.files:
                if k.startswith("g0/"):
                    assert relerr(te.g[k[3:]], g[k]) < 5e-5, k
    sd = m.state_dict()
    for k in g.files:
        if k.startswith("post/"):
            # Adam normalises rounding noise where the true gradient is ~0 (k-bias, saturated sinusoid columns)
            tol = 2e-3 if k.endswith("qkv.bias") else (2e-4 if k.endswith("time_emb.mlp.0.weight") else 2e-5)
            assert relerr(sd[k[5:]], g[k]) < tol, k
    assert torch.equal(sd["condi_emb.weight"].cpu(), p0["condi_emb.weight"])


@pytest.mark.parametrize("dtype,G,side", [("fp32", 64, "1"), ("bf16", 64, "2"), ("fp32", 50, "2"), ("bf16", 200, "1")])
def test_c_enqueue_equals_python_enqueue(dtype, G, side, monkeypatch):
    """stdiff_dit_train_forward / stdiff_dit_backward (ONE C-ABI enqueue each, the default) against the same launches
    issued one by one from Python (STDIFF_TRAIN_ENQUEUE=py): eps, loss, every gradient and the weights after two
    iterations agree bit for bit (up to the run-to-run variation of the atomically summed gradient norm where it
    enters the clip factor), without and with the weight-gradient fork (side = "2": eager fork), for a gene count
    that takes the transposed-copy path (50) and in both compute types; the bucket callback fires depth + 2 times in
    order."""
    from stdiff_amd.train_engine import TrainEngine
    monkeypatch.setenv("STDIFF_TRAIN_SIDE_STREAM", side)
    H, depth, heads, T, C = 128, 3, 4, 1000, 72
    mask = O.kfold_mask(G, 1)
    st = (O.synth_expression(2 * C, G, 5) * 2 - 1).astype(np.float32)
    gen = torch.Generator().manual_seed(3)
    res = {}
    for mode in ("c", "py"):
        m, _ = build(G, H, depth, heads, 70, dtype)
        te = TrainEngine(m, lr=1e-3, diffusion_step=T, mask=mask)
        assert te.c_enqueue
        te.c_enqueue = mode == "c"
        gen.manual_seed(3)
        out = []
        for i in range(2):
            x = torch.from_numpy(st[i * C:(i + 1) * C]).to(DEV)
            xc = (x * torch.from_numpy(mask).to(DEV)).float()
            noise = torch.randn(C, G, generator=gen).to(DEV)
            tt = torch.randint(1, T, (C,), generator=gen).to(DEV)
            loss, gn = te.step(x, xc, noise=noise, timesteps=tt)
            out.append((float(loss), float(gn), te.acts(C)["eps"].clone(), te.flat_g.clone()))
        out.append(te.flat_p.clone())
        if mode == "c":                      # bucket order of the C backward (what the data-parallel step hooks into)
            seen = []
            a = te.acts(C)
            te.backward(a)                   # on the updated weights: the reference point for the hooked call
            plain = te.flat_g.clone()
            te.backward(a, after_bucket=seen.append)
            assert seen == list(range(depth + 2))
            assert torch.equal(te.flat_g, plain)            # same inputs -> same gradients, bucket hooks or not
        res[mode] = out
    for i in range(2):
        # loss and gradient norm are summed with float atomics (their last bits vary run to run); tensors are exact
        for j in (0, 1):
            assert abs(res["c"][i][j] - res["py"][i][j]) < 1e-5 * res["py"][i][j], (i, j)
        if i == 0 or max(res["py"][0][1], res["c"][0][1]) < 1.0:
            # bit for bit -- the second iteration too unless the first one clipped (the clip factor carries the norm's
            # varying last bits into the weights)
            assert torch.equal(res["c"][i][2], res["py"][i][2]) and torch.equal(res["c"][i][3], res["py"][i][3]), i
        else:
            tol = 1e-5 if dtype == "fp32" else 1e-2
            assert relerr(res["c"][i][2], res["py"][i][2]) < tol and relerr(res["c"][i][3], res["py"][i][3]) < tol, i
    if max(res["py"][0][1], res["c"][0][1], res["py"][1][1], res["c"][1][1]) < 1.0:
        assert torch.equal(res["c"][2], res["py"][2])
    else:                                            # Adam's normalisation amplifies the clip factor's last bits
        assert relerr(res["c"][2], res["py"][2]) < 1e-4


@pytest.mark.parametrize("G", [33, 50])
@pytest.mark.parametrize("use_graph", [False, True])
def test_train_odd_gene_counts_against_oracle(G, use_graph):
    """Gene counts that are not a multiple of 8 (diffusion_impute takes any): the weight gradients of in / cond / out
    layer go through transposed copies whose scratch is preallocated per batch size, so the iteration can be captured
    into a hipGraph.  Two full batches and the partial last one (a second graph), each against the oracle."""
    from stdiff_amd.train_engine import TrainEngine
    H, depth, heads, T = 128, 2, 4, 1000
    m, p = build(G, H, depth, heads, 40 + G, "fp32")
    mask = O.kfold_mask(G, 0)
    st = (O.synth_expression(168, G, 3) * 2 - 1).astype(np.float32)
    tb = O.make_tables(T, schedule="cosine")
    pr, state = {k: v.clone() for k, v in p.items()}, {}
    te = TrainEngine(m, lr=1e-5, diffusion_step=T, mask=mask)      # small steps: Adam's sign-like first updates amplify
    gen = torch.Generator().manual_seed(G)                          # rounding noise where |g| ~ 0 by at most lr
    for lo, hi in ((0, 64), (64, 128), (128, 168), (0, 64)):
        C = hi - lo
        x = torch.from_numpy(st[lo:hi])
        xc = torch.from_numpy(st[lo:hi] * mask).float()
        noise = torch.randn(C, G, generator=gen)
        tt = torch.randint(1, T, (C,), generator=gen)
        loss_r, gn_r, raw = O.train_step(pr, tb, state, x, xc, noise, tt, mask, 1e-5, heads, depth)
        loss, gn = te.step(x.to(DEV), xc.to(DEV), noise=noise.to(DEV), timesteps=tt.to(DEV), use_graph=use_graph)
        assert abs(float(loss) - float(loss_r)) < 1e-5 * max(1.0, float(loss_r)), (lo, float(loss), float(loss_r))
        assert abs(float(gn) - float(gn_r)) < 2e-4 * float(gn_r), (lo, float(gn), float(gn_r))
        gscale = max(float(g.abs().max()) for g in raw.values())
        for n in ("in_layer.0.weight", "cond_layer.0.weight", "out_layer.linear.weight", "out_layer.linear.bias"):
            assert float((te.g[n].cpu() - raw[n]).abs().max()) < 1e-4 * gscale, (lo, n)
    if use_graph:
        assert len(te._graphs) == 2 and all("a" in sets[0] for sets in te._graphs.values())   # one graph per batch size


def test_autograd_bridge_matches_oracle():
    """model(x,t,y) under autograd: user-side loss.backward() fills p.grad like the oracle's autograd."""
    G, H, depth, heads, C = 48, 128, 2, 4, 40
    m, p = build(G, H, depth, heads, 21, "fp32")
    m.train()
    x, y = rnd(C, G, seed=1), rnd(C, G, seed=2)
    t = torch.randint(0, 1000, (C,), generator=torch.Generator().manual_seed(3))
    w = rnd(C, G, seed=4)
    out = m(x, t, y)
    (out * w.to(DEV)).sum().backward()
    leaves = {k: v.clone().requires_grad_(True) for k, v in p.items() if k != "condi_emb.weight"}
    q = dict(p)
    q.update(leaves)
    ref = O.dit_forward(q, x, t, y, heads, depth)
    (ref * w).sum().backward()
    assert relerr(out, ref) < 1e-5
    for n, prm in m.named_parameters():
        if n == "condi_emb.weight":
            assert prm.grad is None
        else:
            assert relerr(prm.grad, leaves[n].grad) < 1e-4, n


def test_autograd_two_forwards_before_backward():
    """loss = f(model(x1,t,y1)) + f(model(x2,t,y2)); loss.backward(): the second forward of the same batch size must
    not touch the first one's saved inputs / activations (each pending forward owns a buffer set)."""
    G, H, depth, heads, C = 48, 128, 2, 4, 40
    m, p = build(G, H, depth, heads, 22, "fp32")
    m.train()
    x1, y1, x2, y2 = rnd(C, G, seed=1), rnd(C, G, seed=2), rnd(C, G, seed=5), rnd(C, G, seed=6)
    t1 = torch.randint(0, 1000, (C,), generator=torch.Generator().manual_seed(3))
    t2 = torch.randint(0, 1000, (C,), generator=torch.Generator().manual_seed(7))
    w1, w2 = rnd(C, G, seed=4), rnd(C, G, seed=8)
    o1 = m(x1, t1, y1)
    o2 = m(x2, t2, y2)
    ((o1 * w1.to(DEV)).sum() + (o2 * w2.to(DEV)).sum()).backward()
    leaves = {k: v.clone().requires_grad_(True) for k, v in p.items() if k != "condi_emb.weight"}
    q = dict(p)
    q.update(leaves)
    r1 = O.dit_forward(q, x1, t1, y1, heads, depth)
    r2 = O.dit_forward(q, x2, t2, y2, heads, depth)
    ((r1 * w1).sum() + (r2 * w2).sum()).backward()
    assert relerr(o1, r1) < 1e-5 and relerr(o2, r2) < 1e-5
    for n, prm in m.named_parameters():
        if n != "condi_emb.weight":
            assert relerr(prm.grad, leaves[n].grad) < 1e-4, n
    # and a third, single forward / backward afterwards reuses a free set
    m.zero_grad()
    o3 = m(x1, t1, y1)
    (o3 * w1.to(DEV)).sum().backward()
    assert relerr(o3, r1) < 1e-5


def test_train_bf16_tracks_fp32(golden_dir):
    from stdiff_amd.train_engine import TrainEngine
    g = np.load(os.path.join(golden_dir, "g6_train.npz"))
    C, G, H, depth, heads, seed, T = [int(v) for v in g["cfg"]]
    m, _ = build(G, H, depth, heads, seed, "bf16")
    te = TrainEngine(m, lr=float(g["lr"][0]), diffusion_step=T, mask=g["mask"])
    x, xc = torch.from_numpy(g["x"]).to(DEV), torch.from_numpy(g["x_cond"]).to(DEV)
    for i in range(3):
        loss, gn = te.step(x[i * C:(i + 1) * C].contiguous(), xc[i * C:(i + 1) * C].contiguous(),
                           noise=torch.from_numpy(g["noise"][i]).to(DEV),
                           timesteps=torch.from_numpy(g["timesteps"][i]).to(DEV))
        print(f"bf16 step {i}: loss {float(loss):.5f} (ref {g['loss'][i]:.5f}) gnorm {float(gn):.4f} (ref {g['gnorm'][i]:.4f})")
        assert abs(float(loss) - g["loss"][i]) < 2e-2 * g["loss"][i]
        assert abs(float(gn) - g["gnorm"][i]) < 5e-2 * g["gnorm"][i]


def test_normal_train_stdiff_reduces_loss():
    """The reference call signature end to end: loss goes down on a tiny synthetic problem; returns None."""
    from stdiff_amd.model.stDiff_model import DiT_stDiff
    from stdiff_amd.model.stDiff_train import normal_train_stDiff
    from stdiff_amd.process.data import get_data_loader
    from stdiff_amd.train_engine import TrainEngine
    G, N = 40, 200
    torch.manual_seed(1202)
    m = DiT_stDiff(G, 128, 2, 'dit', 4, 6)
    mask = O.kfold_mask(G, 0)
    seq = O.synth_expression(N, G, 3)
    dl = get_data_loader(seq * 2 - 1, (seq * mask) * 2 - 1, batch_size=64, is_shuffle=True)     # last batch partial (8)
    w0 = {k: v.clone() for k, v in m.state_dict().items()}
    r = normal_train_stDiff(m, dl, lr=1e-3, num_epoch=15, diffusion_step=100, device=torch.device(DEV), is_tqdm=False,
                            mask=mask)
    assert r is None
    sd = m.state_dict()
    assert all(torch.isfinite(v).all() for v in sd.values())
    assert float((sd["out_layer.linear.weight"].cpu() - w0["out_layer.linear.weight"]).abs().max()) > 0
    # evaluate the diffusion loss with a fixed draw before/after
    x = torch.from_numpy(seq[:64] * 2 - 1).to(DEV)
    xc = torch.from_numpy((seq[:64] * mask) * 2 - 1).to(DEV)
    noise, ts = rnd(64, G, seed=9).to(DEV), torch.randint(1, 100, (64,), generator=torch.Generator().manual_seed(1)).to(DEV)

    def eval_loss(model):
        te = TrainEngine(model, lr=0.0, diffusion_step=100, mask=mask)
        loss, _ = te.step(x, xc, noise=noise, timesteps=ts)
        return float(loss)
    after = eval_loss(m)
    fresh = DiT_stDiff(G, 128, 2, 'dit', 4, 6).to(DEV)
    fresh.load_state_dict(w0)
    before = eval_loss(fresh)
    print(f"diffusion loss before {before:.4f} after {after:.4f}")
    assert after < 0.7 * before


# ---------------------------------------------------------------- data-parallel TrainEngine, 2 ranks sharing cuda:0
def _ddp_worker(rank, world, port, golden_dir, q, backend="gloo"):
    import torch.distributed as dist
    import faulthandler
    faulthandler.enable()
    trace = None
    if os.environ.get("STDIFF_TEST_DUMP"):               # stack of a stuck worker after that many seconds
        faulthandler.dump_traceback_later(int(os.environ["STDIFF_TEST_DUMP"]), exit=True)
        trace = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", f"ddp_worker{rank}.log"), "a")

    def stage(msg):
        if trace:
            import time
            trace.write(f"{time.time():.1f} rank {rank}: {msg}\n")
            trace.flush()
    stage("start")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    DEV = f"cuda:{rank}" if backend == "nccl" else "cuda:0"          # RCCL: one GPU per rank; gloo: both on cuda:0
    torch.cuda.set_device(DEV)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(DEV))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from stdiff_amd.train_engine import TrainEngine
        g6 = np.load(os.path.join(golden_dir, "g6_train.npz"))
        g7 = np.load(os.path.join(golden_dir, "g7_ddp.npz"))
        C, G, H, depth, heads, seed, T = [int(v) for v in g6["cfg"]]
        stage("process group up")
        m, _ = build(G, H, depth, heads, seed, "fp32")
        m = m.to(DEV)
        stage("model on device")
        if rank == 1:                                    # rank 0's parameters must win (broadcast at construction)
            with torch.no_grad():
                m.blks[0].mlp.fc1.weight.add_(1.0)
        te = TrainEngine(m, lr=float(g6["lr"][0]), diffusion_step=T, mask=g6["mask"])
        assert te.world == world and te.async_buckets == (backend == "nccl")
        stage("engine built (parameters broadcast)")
        x = torch.from_numpy(g6["x"])[rank * C:(rank + 1) * C].contiguous().to(DEV)
        xc = torch.from_numpy(g6["x_cond"])[rank * C:(rank + 1) * C].contiguous().to(DEV)
        loss, gn = te.step(x, xc, noise=torch.from_numpy(g6["noise"][rank]).to(DEV),
                           timesteps=torch.from_numpy(g6["timesteps"][rank]).to(DEV))
        torch.cuda.synchronize()
        stage("step done")
        assert abs(float(gn) - g7["gnorm"][0]) < 1e-4 * g7["gnorm"][0], (float(gn), g7["gnorm"][0])
        sd = m.state_dict()
        worst = 0.0
        for k in g7.files:
            if k.startswith("post/"):
                # first Adam step: the update is lr * g / (|g| + eps), which amplifies fp32 summation-order noise where |g| is tiny
                tol = 2e-3 if k.endswith("qkv.bias") else (2e-4 if k.endswith("time_emb.mlp.0.weight") else 5e-5)
                e = relerr(sd[k[5:]], g7[k])
                assert e < tol, (k, e)
                worst = max(worst, e)
        q.put((rank, "ok", worst))
    except Exception as e:                                   # noqa: BLE001 -- reported to the parent
        import traceback
        q.put((rank, "fail", traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def _ddp_backends():
    # the RCCL branch (asynchronous stream-ordered buckets) needs one GPU per rank: it runs wherever two are visible
    return ["gloo", "nccl"] if torch.cuda.device_count() >= 2 else ["gloo"]


@pytest.mark.parametrize("backend", _ddp_backends())
def test_ddp_train_step_two_ranks_golden(golden_dir, backend):
    """SURVEY 8(e) training semantics on the GPU path: two processes each run TrainEngine on ITS batch; the bucketed
    all-reduce between backward segments, the 1/W scaling inside clip+AdamW and the update must reproduce golden G7
    (reference gradients of two batches averaged, one optimizer step) on both ranks -- also when rank 1 was built with
    different weights (rank 0's are broadcast at construction, as torch DDP does).  backend "gloo": both ranks on
    cuda:0 (a one-GPU box); backend "nccl" (= RCCL, asynchronous buckets): one GPU per rank, collected only where two
    GPUs are visible."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, golden_dir, q, backend)) for r in range(2)]
    for p_ in procs:
        p_.start()
    res = [q.get(timeout=int(os.environ.get("STDIFF_TEST_DUMP", "120")) + 30) for _ in procs]      # stays under the box's 3-minute silence limit
    for p_ in procs:
        p_.join(60)
    for rank, status, info in res:
        assert status == "ok", f"rank {rank}: {info}"


# ---------------------------------------------------------------- spot-sharded sampling, two ranks
_SHARD_CFG = dict(N=7 * 48 + 20, G=64, H=128, depth=2, heads=4, B=48, T=8, seed=31, noise_seed=77)


def _shard_inputs():
    c = _SHARD_CFG
    mk = O.kfold_mask(c["G"], 0)
    gt = ((O.synth_expression(c["N"], c["G"], 5) * mk) * 2 - 1).astype(np.float32)
    return mk, gt


def _shard_worker(rank, world, port, q, backend):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = f"cuda:{rank}" if backend == "nccl" else "cuda:0"
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from stdiff_amd import parallel as P
        from stdiff_amd.model.stDiff_scheduler import NoiseScheduler
        from stdiff_amd.process.data import get_data_loader
        c = _SHARD_CFG
        m, _ = build(c["G"], c["H"], c["depth"], c["heads"], c["seed"], "fp32")
        m = m.to(dev)
        mk, gt = _shard_inputs()
        uploaded = []
        real = P.loader_condition_rows

        def spy(dl, lo, hi, device):
            uploaded.append((lo, hi))
            return real(dl, lo, hi, device)
        P.loader_condition_rows = spy
        dl = get_data_loader(gt, gt, batch_size=c["B"], is_shuffle=False)              # the reference's host loader
        out = P.sample_stDiff_sharded(m, dl, NoiseScheduler(c["T"], beta_schedule="cosine"), mk, gt, num_step=c["T"],
                                      sample_shape=gt.shape, sample_intermediate=c["T"], seed=c["noise_seed"])
        lo, hi, local = P.sample_stDiff_sharded(m, dl, NoiseScheduler(c["T"], beta_schedule="cosine"), mk, gt,
                                                num_step=c["T"], seed=c["noise_seed"], gather=False)
        # the checksum the bench line carries, through the same object
        ss = P.ShardedSampler(m, NoiseScheduler(c["T"], beta_schedule="cosine"), mk, P.loader_chunk_lens(dl),
                              lambda a, b: (torch.from_numpy(gt[a:b]).to(dev), torch.from_numpy(gt[a:b]).to(dev)),
                              seed=c["noise_seed"])
        ss.init()
        ss.run(list(range(c["T"]))[::-1])
        digest, total = ss.checksum()
        q.put((rank, "ok", dict(out=out, lo=lo, hi=hi, local=local, uploaded=uploaded, digest=digest, total=total)))
    except Exception:                                        # noqa: BLE001 -- reported to the parent
        import traceback
        q.put((rank, "fail", traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("backend", _ddp_backends())
def test_sharded_sampling_two_ranks_equals_single_process(backend):
    """sample_stDiff_sharded on two ranks (gloo: both on cuda:0; nccl where two GPUs are visible) == sample_stDiff in one
    process, BIT FOR BIT: chunk-aligned shards, noise keyed by the global row, one device all_gather of padded shards.
    Each rank uploads only ITS rows, and the cross-rank checksum of the bench line equals the single-process one."""
    import socket
    import torch.multiprocessing as mp
    from stdiff_amd import parallel as P
    from stdiff_amd.model.sample import sample_stDiff
    from stdiff_amd.model.stDiff_scheduler import NoiseScheduler
    from stdiff_amd.process.data import get_data_loader
    c = _SHARD_CFG
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_shard_worker, args=(r, 2, port, q, backend)) for r in range(2)]
    for p_ in procs:
        p_.start()
    res = [q.get(timeout=150) for _ in procs]
    for p_ in procs:
        p_.join(60)
    for rank, status, info in res:
        assert status == "ok", f"rank {rank}: {info}"
    m, _ = build(c["G"], c["H"], c["depth"], c["heads"], c["seed"], "fp32")
    mk, gt = _shard_inputs()
    dl = get_data_loader(gt, gt, batch_size=c["B"], is_shuffle=False)
    ref = sample_stDiff(m, dl, NoiseScheduler(c["T"], beta_schedule="cosine"), mask=mk, gt=gt, device=DEV, num_step=c["T"],
                        sample_shape=gt.shape, is_condi=True, sample_intermediate=c["T"], seed=c["noise_seed"],
                        is_tqdm=False)
    lens = P.loader_chunk_lens(dl)
    assert lens == [48] * 7 + [20]
    one = P.ShardedSampler(m, NoiseScheduler(c["T"], beta_schedule="cosine"), mk, lens,
                           lambda a, b: (torch.from_numpy(gt[a:b]).to(DEV), torch.from_numpy(gt[a:b]).to(DEV)),
                           seed=c["noise_seed"])
    one.init()
    one.run(list(range(c["T"]))[::-1])
    digest1, total1 = one.checksum()
    assert np.array_equal(one.x.cpu().numpy(), ref)
    for rank, _, r in sorted(res):
        lo, hi, _ = P.shard_chunks(lens, rank, 2)
        assert (r["lo"], r["hi"]) == (lo, hi) and np.array_equal(r["local"], ref[lo:hi])
        assert np.array_equal(r["out"], ref), rank                       # gathered result, every rank
        assert set(r["uploaded"]) == {(lo, hi)}, r["uploaded"]            # only this rank's rows went to the GPU
        assert r["digest"] == digest1 and abs(r["total"] - total1) <= 1e-9 * abs(total1)


# ---------------------------------------------------------------- fold-parallel driver, two ranks
def _fold_worker(rank, world, port, golden_dir, ckdir, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device("cuda:0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from stdiff_amd import driver
        g = np.load(os.path.join(golden_dir, "g8_driver.npz"))
        n_sc, n_st, G, H, depth, heads, B, T, num_epoch, n_splits, rand = [int(v) for v in g["cfg"]]
        ran = []
        real = driver.sample_stDiff

        def spy(*a, **k):
            ran.append(1)
            return real(*a, **k)
        driver.sample_stDiff = spy
        impu = driver.diffusion_impute_fold_parallel(
            g["seq"], g["st"], lr=float(g["lr"][0]), depth=depth, num_epoch=num_epoch, diffusion_step=T, batch_size=B,
            hidden_size=H, head=heads, n_splits=n_splits, rand=rand, device=torch.device("cuda:0"), save_dir=ckdir,
            sample_noise=lambda idx: (g[f"x_T{idx}"], g[f"z{idx}"]))
        q.put((rank, "ok", dict(impu=impu, folds_run=len(ran))))
    except Exception:                                        # noqa: BLE001 -- reported to the parent
        import traceback
        q.put((rank, "fail", traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_fold_parallel_driver_two_ranks(golden_dir, tmp_path):
    """The five folds of golden G8 spread over two ranks (3 + 2 folds; both on cuda:0 over gloo), merged by one
    all-reduce: every rank returns the reference's `impu` (<= 2e-5, as the single-process driver test) and the two
    ranks' matrices are bit-equal to each other and to the single-process driver's."""
    import socket
    import torch.multiprocessing as mp
    from stdiff_amd.driver import diffusion_impute
    g = np.load(os.path.join(golden_dir, "g8_driver.npz"))
    n_sc, n_st, G, H, depth, heads, B, T, num_epoch, n_splits, rand = [int(v) for v in g["cfg"]]
    for idx in range(1, n_splits + 1):
        sd = {k[len(f"sd{idx}/"):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(f"sd{idx}/")}
        torch.save(sd, os.path.join(tmp_path, "stDiff%d.pt" % idx))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_fold_worker, args=(r, 2, port, golden_dir, str(tmp_path), q)) for r in range(2)]
    for p_ in procs:
        p_.start()
    res = [q.get(timeout=150) for _ in procs]
    for p_ in procs:
        p_.join(60)
    for rank, status, info in res:
        assert status == "ok", f"rank {rank}: {info}"
    one = diffusion_impute(g["seq"], g["st"], lr=float(g["lr"][0]), depth=depth, num_epoch=num_epoch, diffusion_step=T,
                           batch_size=B, hidden_size=H, head=heads, n_splits=n_splits, rand=rand,
                           device=torch.device(DEV), save_dir=str(tmp_path),
                           sample_noise=lambda idx: (g[f"x_T{idx}"], g[f"z{idx}"]))
    by_rank = {rank: info for rank, _, info in res}
    assert by_rank[0]["folds_run"] == 3 and by_rank[1]["folds_run"] == 2
    for rank in (0, 1):
        assert np.array_equal(by_rank[rank]["impu"], one), rank
        assert float(np.abs(by_rank[rank]["impu"] - g["impu"]).max()) < 2e-5


def test_rccl_bucket_path_on_one_rank(golden_dir, monkeypatch):
    """The data-parallel step's RCCL branch (backend "nccl": every gradient bucket all-reduced asynchronously on RCCL's
    stream as the backward pass completes it, handles waited before the gradient norm) on the ONE GPU this pool's boxes
    have: a one-rank RCCL group, buckets forced (STDIFF_DDP_FORCE_BUCKETS=1; each all-reduce is an identity), three G6
    iterations.  Transport between GPUs is not exercised; the stream ordering, the bucket hooks of the C-ABI backward
    and the communicator calls are."""
    import socket
    import torch.distributed as dist
    from stdiff_amd.train_engine import TrainEngine
    monkeypatch.setenv("STDIFF_DDP_FORCE_BUCKETS", "1")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device(DEV))
    try:
        g = np.load(os.path.join(golden_dir, "g6_train.npz"))
        C, G, H, depth, heads, seed, T = [int(v) for v in g["cfg"]]
        m, _ = build(G, H, depth, heads, seed, "fp32")
        te = TrainEngine(m, lr=float(g["lr"][0]), diffusion_step=T, mask=g["mask"])
        assert te.world == 1 and te.async_buckets and te.force_buckets
        calls = []
        real = dist.all_reduce

        def spy(t, *a, **kw):
            calls.append((t.numel(), kw.get("async_op", False)))
            return real(t, *a, **kw)
        monkeypatch.setattr(te.dist, "all_reduce", spy)
        x, xc = torch.from_numpy(g["x"]).to(DEV), torch.from_numpy(g["x_cond"]).to(DEV)
        for i in range(3):
            loss, gn = te.step(x[i * C:(i + 1) * C].contiguous(), xc[i * C:(i + 1) * C].contiguous(),
                               noise=torch.from_numpy(g["noise"][i]).to(DEV),
                               timesteps=torch.from_numpy(g["timesteps"][i]).to(DEV))
            assert abs(float(loss) - g["loss"][i]) < 1e-5 * max(1.0, g["loss"][i]), i
            assert abs(float(gn) - g["gnorm"][i]) < 1e-4 * g["gnorm"][i], i
        assert len(calls) == 3 * (depth + 2) and all(a for _, a in calls)
        assert sum(n for n, _ in calls[:depth + 2]) == te.total          # the buckets cover the flat gradient once
        sd = m.state_dict()
        for k in g.files:
            if k.startswith("post/"):
                tol = 2e-3 if k.endswith("qkv.bias") else (2e-4 if k.endswith("time_emb.mlp.0.weight") else 2e-5)
                assert relerr(sd[k[5:]], g[k]) < tol, k
    finally:
        dist.destroy_process_group()


def test_fp16_sampling_tolerance_on_trained_weights():
    """Train from the reference initialisation (zero adaLN / output layers, Xavier elsewhere) for 60 epochs, then run
    the denoiser with fp16 operands on the TRAINED weights: the per-call error against the oracle stays inside 1e-3
    of the output scale (activation ranges of a trained model, not of re-randomised weights), and the model refuses
    to train in fp16."""
    from stdiff_amd.model.stDiff_model import DiT_stDiff
    from stdiff_amd.model.stDiff_train import normal_train_stDiff
    from stdiff_amd.process.data import get_data_loader
    from stdiff_amd.train_engine import TrainEngine
    G, H, depth, heads, N = 64, 256, 3, 4, 512
    torch.manual_seed(1202)
    m = DiT_stDiff(G, H, depth, 'dit', heads, 6, compute_dtype="bf16")
    mask = O.kfold_mask(G, 0)
    seq = O.synth_expression(N, G, 3)
    dl = get_data_loader(seq * 2 - 1, (seq * mask) * 2 - 1, batch_size=128, is_shuffle=True)
    normal_train_stDiff(m, dl, lr=1e-3, num_epoch=60, diffusion_step=1000, device=torch.device(DEV), is_tqdm=False,
                        mask=mask)
    p = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    assert float(p["out_layer.linear.weight"].abs().max()) > 0          # the zero-initialised layers have moved
    x, y = rnd(128, G, seed=4), torch.from_numpy((seq[:128] * mask) * 2 - 1)
    worst = 0.0
    for t in (999, 500, 20):
        ts = torch.full((128,), t, dtype=torch.long)
        ref = O.dit_forward(p, x, ts, y, heads, depth)
        for dtype, tol in (("fp32", 1e-5), ("fp16", 1e-3)):
            m.set_compute_dtype(dtype)
            with torch.no_grad():
                e = relerr(m(x, ts, y), ref)
            print(f"trained weights, t={t}, {dtype}: rel err {e:.3e}")
            assert e < tol
            worst = max(worst, e)
    with pytest.raises(ValueError, match="fp16"):
        TrainEngine(m, lr=1e-4, diffusion_step=1000, mask=mask)


@pytest.mark.parametrize("dt", ["fp32", "bf16"])
def test_deferred_split_consumers_equal_the_reduce_launch(dt):
    """stdiff_gemm with `defer` leaves a split-K result as partial products; the three consumers (LayerNorm forward with
    the producer's bias + gated residual, LayerNorm backward with and without the gate product, attention backward) must
    reproduce the unfused sequence -- reduce launch, then the plain entry point -- bit for bit."""
    from stdiff_amd import ops, _lib as L
    tdt = torch.float32 if dt == "fp32" else torch.bfloat16
    gen = torch.Generator().manual_seed(11)
    C_, H, K, heads = 200, 256, 2048, 4
    r = lambda *s, sc=1.0: (torch.randn(*s, generator=gen) * sc).to(DEV)
    A, W, bias = r(C_, K).to(tdt), r(H, K, sc=K ** -0.5).to(tdt), r(H)
    mod = r(C_, 3 * H).to(tdt)
    res = r(C_, H)
    ws = torch.empty(8 << 20, device=DEV, dtype=torch.float32)
    # ---- forward: x = res + gate * (A W^T + bias), LayerNorm-modulate of x
    x0, v0 = torch.empty(C_, H, device=DEV), torch.empty(C_, H, device=DEV, dtype=tdt)
    ops.gemm(A, W, bias, out=x0, epi=L.EPI_GATE_RES, gate=mod[:, 2 * H:], res=res, out2=v0, workspace=ws)
    m0, s0 = torch.empty(C_, device=DEV), torch.empty(C_, device=DEV)
    y0 = ops.ln_modulate(x0, mod[:, :H], mod[:, H:2 * H], mean=m0, rstd=s0)
    sp = L.SplitResult()
    x1, v1 = torch.zeros(C_, H, device=DEV), torch.zeros(C_, H, device=DEV, dtype=tdt)
    ops.gemm(A, W, bias, out=x1, epi=L.EPI_GATE_RES, gate=mod[:, 2 * H:], res=res, out2=v1, workspace=ws, defer=sp)
    assert sp.slices >= 3 and sp.ldws >= H
    m1, s1 = torch.empty(C_, device=DEV), torch.empty(C_, device=DEV)
    y1 = ops.ln_modulate(x1, mod[:, :H], mod[:, H:2 * H], mean=m1, rstd=s1, split=sp, bias=bias, gate=mod[:, 2 * H:],
                         res=res, v_out=v1)
    for a_, b_ in ((x0, x1), (v0, v1), (y0, y1), (m0, m1), (s0, s1)):
        assert torch.equal(a_, b_)
    # without a workspace nothing splits: slices == 0 and the GEMM is complete
    sp2 = L.SplitResult()
    x2 = ops.gemm(A, W, bias, out=torch.empty(C_, H, device=DEV), defer=sp2)
    assert sp2.slices == 0 and relerr(x2, A.float() @ W.float().T + bias) < 1e-4
    # ---- LayerNorm backward of dy = dY W (K-major W), with and without the gate product
    dY, Wk = r(C_, K).to(tdt), r(K, H, sc=K ** -0.5).to(tdt)
    scale, gate, branch = r(C_, H, sc=0.3).to(tdt), r(C_, H).to(tdt), r(C_, H).to(tdt)
    for with_gate in (True, False):
        outs = []
        for fused in (False, True):
            sp = L.SplitResult() if fused else None
            dy = torch.zeros(C_, H, device=DEV, dtype=tdt)
            ops.gemm(dY, Wk, None, out=dy, w_kmajor=True, workspace=ws, defer=sp)
            dx = torch.ones(C_, H, device=DEV)
            dsh, dsc = torch.zeros(C_, H, device=DEV, dtype=tdt), torch.zeros(C_, H, device=DEV, dtype=tdt)
            db, dg = torch.zeros(C_, H, device=DEV, dtype=tdt), torch.zeros(C_, H, device=DEV, dtype=tdt)
            kw = dict(gate=gate, branch=branch, dbranch=db, dgate=dg) if with_gate else {}
            ops.ln_modulate_bwd(dy, x0, m0, s0, scale, dx, dsh, dsc, accumulate=True, split=sp, **kw)
            assert not fused or sp.slices >= 3
            outs.append((dx, dsh, dsc, db, dg))
        for a_, b_ in zip(*outs):
            assert torch.equal(a_, b_)
    # ---- attention backward with dout = dY W
    Hh = heads * 64
    qkv, Wp = r(C_, 3 * Hh).to(tdt), r(K, Hh, sc=K ** -0.5).to(tdt)
    cu = torch.tensor([0, 120, C_], dtype=torch.int32, device=DEV)
    lse = torch.empty(C_, heads, device=DEV)
    o = ops.attention_fwd(qkv, cu, 120, heads, lse=lse)
    outs = []
    for fused in (False, True):
        sp = L.SplitResult() if fused else None
        do = torch.zeros(C_, Hh, device=DEV, dtype=tdt)
        ops.gemm(dY, Wp, None, out=do, w_kmajor=True, workspace=ws, defer=sp)
        dq = ops.attention_bwd(qkv, o, do, lse, cu, 120, heads, split=sp)
        assert not fused or sp.slices >= 3
        outs.append((do, dq))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("dtype,use_graph,enqueue", [("fp32", False, "c"), ("bf16", True, "c"), ("bf16", False, "py"),
                                                      ("fp32", True, "c")])
def test_lazy_update_equals_eager_update(dtype, use_graph, enqueue, monkeypatch):
    """TrainEngine(lazy_update=True): the clip + AdamW update of iteration i is issued by iteration i + 1 on a stream of
    its own, one range of the flat buffers per layer, with the forward waiting per layer (stdiff_adamw_range,
    param_ready).  Same arithmetic: after finish() parameters, moments and every loss equal the eager engine's; before
    finish() the parameters are those of the previous update."""
    from stdiff_amd.train_engine import TrainEngine
    monkeypatch.setenv("STDIFF_TRAIN_ENQUEUE", enqueue)
    G, H, depth, heads, T, C = 72, 128, 3, 4, 1000, 96
    mask = O.kfold_mask(G, 1)
    st = (O.synth_expression(C, G, 9) * 2 - 1).astype(np.float32)
    x = torch.from_numpy(st).to(DEV)
    xc = (x * torch.from_numpy(mask).to(DEV)).float()
    res = {}
    for lazy in (False, True):
        m, _ = build(G, H, depth, heads, 71, dtype)
        te = TrainEngine(m, lr=1e-3, diffusion_step=T, mask=mask, lazy_update=lazy)
        assert te.lazy == lazy
        gen = torch.Generator().manual_seed(5)
        losses = []
        for i in range(5):
            noise = torch.randn(C, G, generator=gen).to(DEV)
            tt = torch.randint(1, T, (C,), generator=gen).to(DEV)
            loss, _ = te.step(x, xc, noise=noise, timesteps=tt, use_graph=use_graph)
            losses.append(float(loss))
            if lazy and i == 3:
                before = te.flat_p.clone()
        if lazy:
            assert te._pending
            stale = te.flat_p.clone()
            te.finish()
            assert not te._pending and not torch.equal(stale, te.flat_p)
            te.finish()                                   # idempotent
        res[lazy] = (losses, te.flat_p.clone(), te.m.clone(), te.v.clone(), int(te.step_dev))
    assert res[True][4] == res[False][4] == 5
    # losses: summed with float atomics (last bits vary run to run); tensors: exact unless a step clipped (the clip
    # factor carries the norm's varying last bits)
    for a_, b_ in zip(res[True][0], res[False][0]):
        assert abs(a_ - b_) < 1e-5 * abs(b_)
    tol = 1e-4 if dtype == "fp32" else 2e-2
    for k in (1, 2, 3):
        assert relerr(res[True][k], res[False][k]) < tol, k


@pytest.mark.parametrize("dtype", ["bf16", "fp32"])
def test_training_iterations_repeat_bit_for_bit(dtype):
    """With the squared gradient norm taken from the weight-gradient accumulators (stdiff_dw_group_args.sq_parts, summed by
    stdiff_sum_parts in a fixed order) nothing in an iteration depends on the order in which workgroups finish: two runs
    from the same seeds give the same parameters, moments and gradient norms, bit for bit (the loss VALUE is still summed
    with float atomics and is compared loosely)."""
    from stdiff_amd.train_engine import TrainEngine
    G, H, depth, heads, T, C = 72, 128, 3, 4, 1000, 96
    mask = O.kfold_mask(G, 1)
    x = torch.from_numpy((O.synth_expression(C, G, 9) * 2 - 1).astype(np.float32)).to(DEV)
    xc = (x * torch.from_numpy(mask).to(DEV)).float()
    runs = []
    for _ in range(2):
        m, _ = build(G, H, depth, heads, 71, dtype)
        te = TrainEngine(m, lr=1e-3, diffusion_step=T, mask=mask, lazy_update=True)
        assert te.fold_norm and te.norm_in_group
        gen = torch.Generator().manual_seed(5)
        norms, losses = [], []
        for i in range(6):
            noise = torch.randn(C, G, generator=gen).to(DEV)
            tt = torch.randint(1, T, (C,), generator=gen).to(DEV)
            loss, _ = te.step(x, xc, noise=noise, timesteps=tt, use_graph=True)
            losses.append(float(loss))
            norms.append(float(te.sumsq))
        te.finish()
        runs.append((te.flat_p.clone(), te.m.clone(), te.v.clone(), norms, losses))
    a, b = runs
    assert a[3] == b[3] and all(n > 0 for n in a[3])
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    assert all(abs(p - q) < 1e-5 * abs(q) for p, q in zip(a[4], b[4]))


@pytest.mark.parametrize("dtype", ["bf16", "fp32"])
def test_unjoined_steps_equal_joined_steps(dtype):
    """step(join=False): the caller's stream does not wait for the iteration; the next step's inputs are staged on a
    stream of their own, into the other of two staging sets (two captured graphs), while the iteration runs.  Same
    kernels on the same data in the same order: parameters, moments and gradient norms equal the joined engine's bit for
    bit, the loss sum kept with accumulate() equals the sum of the joined losses; inputs are dropped right after each
    call (record_stream keeps them alive for the staging stream)."""
    from stdiff_amd.train_engine import TrainEngine
    G, H, depth, heads, T, C = 72, 128, 3, 4, 1000, 96
    mask = O.kfold_mask(G, 1)
    base = torch.from_numpy((O.synth_expression(C, G, 9) * 2 - 1).astype(np.float32)).to(DEV)
    mk = torch.from_numpy(mask).to(DEV)
    res = {}
    for join in (True, False):
        m, _ = build(G, H, depth, heads, 71, dtype)
        te = TrainEngine(m, lr=1e-3, diffusion_step=T, mask=mask, lazy_update=True)
        gen = torch.Generator().manual_seed(5)
        total = torch.zeros(1, device=DEV)
        norms = []
        for i in range(7):
            x = base * (1.0 + 0.01 * i)                       # a fresh tensor per step, dropped after the call
            xc = (x * mk).float()
            noise = torch.randn(C, G, generator=gen).to(DEV)
            tt = torch.randint(1, T, (C,), generator=gen).to(DEV)
            loss, _ = te.step(x, xc, noise=noise, timesteps=tt, use_graph=True, join=join)
            te.accumulate(total, loss)
            del x, xc, noise, tt
            if join:
                norms.append(float(te.sumsq))
        te.finish()
        assert not te._pending
        res[join] = (te.flat_p.clone(), te.m.clone(), te.v.clone(), float(total), float(te.scal[3]))
        if not join:
            assert all(s is not None and s["graph"] is not None for s in te._graphs[C])      # both sets were used
    a, b = res[True], res[False]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    assert a[4] == b[4] and abs(a[3] - b[3]) < 1e-5 * abs(a[3])


@pytest.mark.parametrize("dtype,enqueue", [("bf16", "c"), ("fp32", "c"), ("bf16", "py")])
def test_update_in_pieces_equals_whole_block_ranges(dtype, enqueue, monkeypatch):
    """The lazy update with the last blocks' ranges cut into four pieces in the order the forward uses them (adaLN
    projection, qkv + proj, fc1, fc2; one param_ready slot each, the default for wide models) against whole-block
    ranges: every element goes through the same arithmetic once, so parameters and moments are equal bit for bit, and
    the pieces tile each block's range exactly."""
    from stdiff_amd.train_engine import TrainEngine
    monkeypatch.setenv("STDIFF_TRAIN_ENQUEUE", enqueue)
    G, H, depth, heads, T, C = 72, 128, 3, 4, 1000, 96
    mask = O.kfold_mask(G, 1)
    x = torch.from_numpy((O.synth_expression(C, G, 9) * 2 - 1).astype(np.float32)).to(DEV)
    xc = (x * torch.from_numpy(mask).to(DEV)).float()
    res = {}
    for pieces in ("0", "2", "3"):
        monkeypatch.setenv("STDIFF_TRAIN_UPDATE_PIECES", pieces)
        m, _ = build(G, H, depth, heads, 71, dtype)
        te = TrainEngine(m, lr=1e-3, diffusion_step=T, mask=mask, lazy_update=True)
        assert te.piece_blocks == int(pieces) and len(te.update_ranges) == depth + 2 + 3 * int(pieces)
        covered = sorted(te.update_ranges)
        assert covered[0][0] == 0 and covered[-1][1] == te.total
        assert all(a[1] == b[0] for a, b in zip(covered, covered[1:]))          # the ranges tile the flat buffer
        gen = torch.Generator().manual_seed(5)
        for i in range(5):
            noise = torch.randn(C, G, generator=gen).to(DEV)
            tt = torch.randint(1, T, (C,), generator=gen).to(DEV)
            te.step(x, xc, noise=noise, timesteps=tt, use_graph=(enqueue == "c"))
        te.finish()
        res[pieces] = (te.flat_p.clone(), te.m.clone(), te.v.clone())
    for pieces in ("2", "3"):
        for a_, b_ in zip(res["0"], res[pieces]):
            assert torch.equal(a_, b_), pieces
