"""Training side of the hot path: DiT_stDiff forward with saved activations, the hand-written backward, the masked
diffusion loss and the fused clip+AdamW step of normal_train_stDiff (stDiff_train.py:56-81), all through
libstdiff_hip.so.  Under data parallelism the flat gradient buffer is all-reduced with RCCL (torch.distributed
backend "nccl"), one bucket per DiT block, overlapped with the rest of the backward pass."""
import os

import torch

from . import _lib as L
from . import ops

_NO_GRAD = ("condi_emb.weight",)      # unused by forward (stDiff_model.py:184,236): no gradient, skipped by clip/AdamW


def flat_layout(named_sizes):
    """Offsets of each tensor in the flat fp32 buffer and its length.  Every tensor starts on a multiple of 8 elements:
    16-byte aligned in the fp32 buffers AND in the bf16 operand mirror that shares these offsets."""
    offs, total = {}, 0
    for n, numel in named_sizes:
        offs[n] = total
        total += (numel + 7) // 8 * 8
    return offs, total


def bucket_ranges(offs, total, depth):
    """Contiguous [lo, hi) ranges of the flat gradient in the order backward completes them: final layer,
    block depth-1 .. block 0, embedding layers.  These are the RCCL all-reduce buckets of data-parallel training."""
    out0 = offs["out_layer.linear.weight"]
    r = [(out0, total)]
    for k in range(depth - 1, -1, -1):
        lo = offs[f"blks.{k}.attn.qkv.weight"]
        hi = offs[f"blks.{k + 1}.attn.qkv.weight"] if k + 1 < depth else out0
        r.append((lo, hi))
    r.append((0, offs["blks.0.attn.qkv.weight"]))
    return r


class TrainCore:
    """Forward with saved activations + backward into ONE flat fp32 gradient buffer."""

    def __init__(self, model):
        self.model = model
        params = dict(model.named_parameters())
        dev = params["in_layer.0.weight"].device
        if dev.type != "cuda":
            raise RuntimeError("stdiff_amd runs on MI355X only: move the model with .to('cuda') (no CPU fallback)")
        self.device = dev
        from .engine import F16, parse_dtype
        if parse_dtype(model.compute_dtype) == F16:
            raise ValueError("compute_dtype 'fp16' is a sampling format (its gradients would need loss scaling): "
                             "train in 'bf16' or 'fp32'")
        self.names = [n for n in params if n not in _NO_GRAD]
        # ---- flat parameter / gradient / moment buffers (each tensor starts on a 16-byte boundary)
        offs, total = flat_layout([(n, params[n].numel()) for n in self.names])
        self.offs, self.total = offs, total
        flat = torch.zeros(total, device=dev, dtype=torch.float32)
        with torch.no_grad():
            for n in self.names:
                p = params[n]
                v = flat[offs[n]:offs[n] + p.numel()].view_as(p)
                v.copy_(p.detach())
                p.data = v
        self.flat_p = flat
        self.flat_g = torch.zeros_like(flat)
        self.g = {n: self.flat_g[offs[n]:offs[n] + params[n].numel()].view_as(params[n]) for n in self.names}
        self.eng = model.engine()          # packs the (now flat-backed) parameters
        self.pw = self.eng.pw
        pw = self.pw
        self.G, self.Gp, self.H, self.depth, self.heads = pw.G, pw.Gp, pw.H, pw.depth, pw.heads
        self.tdt = pw.tdt
        # ---- operand copies kept current by the optimizer kernel instead of a repack pass per weight: bf16 compute
        # reads a flat bf16 mirror (same offsets) that clip_adamw writes; fp32 compute reads the master weights in
        # place.  Only the gene-padded in/cond weights ([H, Gp] != [H, G]) still go through pack_rows.
        self.mirror = None
        views = {}
        if self.tdt == torch.bfloat16:
            self.mirror = flat.to(torch.bfloat16)
        for n in pw.w:
            if pw.w[n].shape == params[n].shape:
                src = self.mirror if self.mirror is not None else flat
                views[n] = src[offs[n]:offs[n] + params[n].numel()].view_as(params[n])
        pw.adopt(views)
        self.mirrored = tuple(views)
        self._acts = {}
        # split-K scratch shared by every GEMM of the step (one stream, so uses never overlap): batch-sized outputs
        # have 32-200 tiles, too few to fill 256 CUs unless the long K axes are cut too
        self.ws_splitk = torch.empty(16 << 20, device=dev, dtype=torch.float32)
        # The weight-gradient contractions (dW = dY^T X, db = colsum dY) are leaves of the backward graph.  While an
        # iteration is being CAPTURED into a hipGraph they are issued on a side stream (own split-K scratch), so the
        # graph gets parallel branches next to the dX chain: at 512 rows a GEMM is a ~15 us launch that leaves most
        # of the chip idle.  Only under capture: there the fork/join events are graph edges and cost nothing at
        # replay; issued eagerly, the 4 event calls per layer make the host the limiter (3.55 -> 4.42 ms per step).
        # STDIFF_TRAIN_SIDE_STREAM=0 never forks, =2 forks in eager mode too (tests).
        self.side_mode = int(os.environ.get("STDIFF_TRAIN_SIDE_STREAM", "1"))
        self.ws_splitk_side = None
        self._side, self._side_prev, self._side_on = None, None, False
        # Forward and backward are ONE C-ABI enqueue each (stdiff_dit_train_forward / stdiff_dit_backward);
        # STDIFF_TRAIN_ENQUEUE=py issues the same launches one by one from Python (second implementation, tests)
        self.c_enqueue = os.environ.get("STDIFF_TRAIN_ENQUEUE", "c") != "py"
        # one stdiff_dw_group launch per block for its five weight/bias gradients (STDIFF_TRAIN_GROUPED=0: one job of
        # three launches per linear layer, the older path, kept for A/B timing and as a second implementation)
        self.grouped = os.environ.get("STDIFF_TRAIN_GROUPED", "1") != "0"
        # a split-K GEMM whose consumer is the next launch leaves its partial products to that consumer (LayerNorm
        # forward / backward, attention backward) instead of paying a reduce launch (STDIFF_TRAIN_DEFER=0: A/B timing;
        # results are bit-identical either way)
        self.defer = os.environ.get("STDIFF_TRAIN_DEFER", "1") != "0"
        # the folded gradient norm (backward(sumsq=...)) comes from the group launches' accumulators
        # (STDIFF_TRAIN_NORM_IN_GROUP=0: one stdiff_grad_sumsq pass per bucket beside the chain, A/B timing)
        self.norm_in_group = os.environ.get("STDIFF_TRAIN_NORM_IN_GROUP", "1") != "0"
        self._cw = None
        self._parts = None

    def refresh_weights(self, pack=True):
        """fp32 master weights -> packed MFMA operands (the backward reads the same [out, in] copies K-major)."""
        if pack:
            self.pw.refresh(skip=self.mirrored)

    def bucket_ranges(self):
        return bucket_ranges(self.offs, self.total, self.depth)

    # ------------------------------------------------------------------ buffers for a batch of C rows
    def acts(self, C):
        a = self._acts.get(C)
        if a is not None:
            return a
        H, G, Gp, D, dev, T = self.H, self.G, self.Gp, self.depth, self.device, self.tdt
        Cp = ops.round_up(C, 64)
        f32 = torch.float32

        def z(*s, dt=T):
            return torch.zeros(s, device=dev, dtype=dt)
        a = dict(
            C=C, Cp=Cp, cu=torch.tensor([0, C], dtype=torch.int32, device=dev),
            xp=z(C, Gp), yp=z(C, Gp), e0=z(C, 256), a1=z(C, H), e1=z(C, H), temb=z(C, H, dt=f32), c=z(C, H, dt=f32),
            s=z(C, H), h=[z(C, H, dt=f32) for _ in range(D + 1)], hmid=[z(C, H, dt=f32) for _ in range(D)],
            mod=[z(C, 6 * H) for _ in range(D)], xn1=[z(C, H) for _ in range(D)], xn2=[z(C, H) for _ in range(D)],
            qkv=[z(C, 3 * H) for _ in range(D)], o=[z(C, H) for _ in range(D)], lse=[z(C, self.heads, dt=f32) for _ in range(D)],
            pa=[z(C, H) for _ in range(D)], pm=[z(C, H) for _ in range(D)], upre=[z(C, 4 * H) for _ in range(D)],
            u=[z(C, 4 * H) for _ in range(D)],
            mean1=[z(C, dt=f32) for _ in range(D)], rstd1=[z(C, dt=f32) for _ in range(D)],
            mean2=[z(C, dt=f32) for _ in range(D)], rstd2=[z(C, dt=f32) for _ in range(D)],
            fmod=z(C, 2 * H), xnf=z(C, H), meanf=z(C, dt=f32), rstdf=z(C, dt=f32), eps=z(C, G, dt=f32),
            # backward scratch
            deps=z(C, Gp), dh=z(C, H, dt=f32), dxn=z(C, H), dmod=[z(C, 6 * H), z(C, 6 * H)], dfmod=z(C, 2 * H), dbr=z(C, H),
            du=z(C, 4 * H), dupre=z(C, 4 * H), do=z(C, H), dqkv=z(C, 3 * H), delta=z(C, self.heads, dt=f32),
            ds32=z(C, H, dt=f32), dc=z(C, H, dt=f32), dcT_=z(C, H), de1=z(C, H), da1=z(C, H), dhT_=z(C, H),
        )
        if self.grouped:
            # second set of the per-block dY buffers (dbr of either branch, dqkv; `du` twins dupre): one grouped
            # weight-gradient launch per block runs while the dX chain is already in the next block
            a["dbr_b"] = [z(C, H) for _ in range(3)]
            a["dqkv_b"] = z(C, 3 * H)
        if G % 8:
            # odd gene counts: the weight-gradient GEMMs of in / cond / out layer go through transposed copies
            # (_dW_enqueue).  Their scratch is made HERE, one pair per stream that may run such a job, so that nothing
            # allocates while an iteration is being captured into a hipGraph (the graph keeps raw addresses).
            a["tr"] = [(z(max(Gp, H), Cp), z(max(Gp, H), Cp)) for _ in range(2)]
        if len(self._acts) > 3:
            # drops the CACHE entries only: captured graphs (st['a']) and autograd contexts (ctx.a) keep references
            # to the buffer sets they recorded, so their addresses stay valid
            self._acts.clear()
        self._acts[C] = a
        return a

    def _gemm(self, *args, side=False, **kw):
        return ops.gemm(*args, workspace=self.ws_splitk_side if side else self.ws_splitk, **kw)

    def _side_begin(self):
        """Decide whether this backward pass forks its weight gradients onto the side stream."""
        self._side_prev, self._side_on = None, False
        if self.side_mode == 0 or (self.side_mode == 1 and not torch.cuda.is_current_stream_capturing()):
            return
        if self._side is None:
            from .engine import aux_stream
            self._side = aux_stream(self.device, "side")      # one per device and process (engine.py, _STREAMS)
            self.ws_splitk_side = torch.empty(16 << 20, device=self.device, dtype=torch.float32)
        self._side_on = True

    def _side_join(self):
        """The caller's stream waits for every weight gradient issued so far."""
        if self._side_on and self._side_prev is not None:
            torch.cuda.current_stream().wait_event(self._side_prev)

    def _lin(self, name):
        return self.pw.w[name + ".weight"], self.pw.b(name + ".bias")

    # ------------------------------------------------------------------ the C-ABI view of weights / activations
    def _c_weights(self):
        """stdiff_train_weights over the packed operands, the biases and the flat gradient buffer (addresses are
        stable: the optimizer and refresh() write in place)."""
        if self._cw is not None:
            return self._cw[0]
        p = L.ptr

        def lin(name):
            return L.TrainLinear(p(self.pw.w[name + ".weight"]), p(self.pw.b(name + ".bias")),
                                 p(self.g[name + ".weight"]), p(self.g[name + ".bias"]))
        blocks = (L.TrainBlock * max(1, self.depth))()
        for k in range(self.depth):
            b = f"blks.{k}."
            blocks[k] = L.TrainBlock(lin(b + "attn.qkv"), lin(b + "attn.proj"), lin(b + "mlp.fc1"), lin(b + "mlp.fc2"),
                                     lin(b + "adaLN_modulation.1"))
        w = L.TrainWeights()
        w.dtype, w.G, w.Gp, w.H, w.heads, w.depth = L.dt_of(self.pw.w["in_layer.0.weight"]), self.G, self.Gp, self.H, \
            self.heads, self.depth
        w.t0, w.t2 = lin("time_emb.mlp.0"), lin("time_emb.mlp.2")
        w.cond, w.in_ = lin("cond_layer.0"), lin("in_layer.0")
        w.fada, w.out = lin("out_layer.adaLN_modulation.1"), lin("out_layer.linear")
        w.blocks, w.freqs = blocks, p(self.pw.freqs)
        self._cw = (w, blocks)
        return w

    def _c_acts(self, a):
        """stdiff_train_acts over the buffer set `a` (cached in it; the side scratch is filled in when a side stream
        first exists)."""
        ca = a.get("_c")
        if ca is None:
            p, D = L.ptr, self.depth
            blocks = (L.TrainBlockActs * max(1, D))()
            for k in range(D):
                blocks[k] = L.TrainBlockActs(*[p(a[n][k]) for n in ("h", "hmid", "mod", "xn1", "xn2", "qkv", "o", "lse", "pa",
                                                                    "pm", "upre", "u", "mean1", "rstd1", "mean2", "rstd2")])
            ca = L.TrainActs()
            ca.C, ca.cu_rows, ca.blocks, ca.h_last = a["C"], p(a["cu"]), blocks, p(a["h"][D])
            for n in ("xp", "yp", "e0", "a1", "e1", "temb", "c", "s", "fmod", "xnf", "meanf", "rstdf", "eps", "deps", "dh",
                      "dxn", "dfmod", "dbr", "du", "dupre", "dqkv", "delta", "ds32", "dc", "de1", "da1"):
                setattr(ca, n, p(a[n]))
            ca.do_, ca.dcT, ca.dhT = p(a["do"]), p(a["dcT_"]), p(a["dhT_"])
            ca.dmod[0], ca.dmod[1] = p(a["dmod"][0]), p(a["dmod"][1])
            if "tr" in a:
                for i in range(2):
                    ca.tr[i][0], ca.tr[i][1] = p(a["tr"][i][0]), p(a["tr"][i][1])
            if "dbr_b" in a:
                for i in range(3):
                    ca.dbr_b[i] = p(a["dbr_b"][i])
                ca.dqkv_b = p(a["dqkv_b"])
            ca.no_defer = 0 if self.defer else 1
            ca.splitk_ws, ca.splitk_bytes = p(self.ws_splitk), self.ws_splitk.numel() * 4
            a["_c"] = ca = (ca, blocks)
        ca = ca[0]
        if self.ws_splitk_side is not None and not ca.splitk_ws_side:
            ca.splitk_ws_side, ca.splitk_side_bytes = L.ptr(self.ws_splitk_side), self.ws_splitk_side.numel() * 4
        return ca

    def _c_side(self):
        """The side stream's handle when this pass forks (same rule as the Python enqueue), else None."""
        self._side_begin()
        on, self._side_on = self._side_on, False
        return self._side.cuda_stream if on else None

    # ------------------------------------------------------------------ forward (stDiff_model.py:231-243)
    def forward(self, a, t, param_ready=None):
        """a['xp'], a['yp'] hold the packed inputs; returns eps fp32 [C, G] (a['eps']).
        param_ready: 4 * depth + 2 slots of recorded torch.cuda.Event or None (stdiff_train_acts.param_ready: embedding
        layers; per block adaLN projection / qkv + proj / fc1 / fc2; final layer): the forward waits for each where those
        parameters are first used (an optimizer updating them on another stream)."""
        if self.c_enqueue:
            side = self._c_side()
            ca = self._c_acts(a)
            if param_ready is not None:
                arr = (L.vp * len(param_ready))(*[e.cuda_event if e is not None else None for e in param_ready])
                ca.param_ready = L.C.cast(arr, L.C.POINTER(L.vp))
            try:
                L.check(L.lib().stdiff_dit_train_forward(self._c_weights(), ca, L.ptr(t), side, L.stream()))
            finally:
                ca.param_ready = L.C.cast(None, L.C.POINTER(L.vp))
            return a["eps"]
        H, D = self.H, self.depth
        g = self._gemm

        def params_ready(i, st=None):
            if param_ready is not None and param_ready[i] is not None:
                (st or torch.cuda.current_stream()).wait_event(param_ready[i])
        ops.timestep_embedding(t, self.pw.freqs, out=a["e0"])
        params_ready(0)
        w, b = self._lin("time_emb.mlp.0")
        g(a["e0"], w, b, out=a["e1"], epi=L.EPI_SILU, out2=a["a1"])
        w, b = self._lin("time_emb.mlp.2")
        g(a["e1"], w, b, out=a["temb"])
        w, b = self._lin("cond_layer.0")
        g(a["yp"], w, b, out=a["c"], epi=L.EPI_ADD, res=a["temb"])
        ops.silu_cond(a["c"], out=a["s"])
        # every adaLN projection depends on s only: under graph capture they go to the side stream as one chain, and
        # the block chain waits for mod[k] where it first needs it
        self._side_begin()
        mod_ready = None
        if self._side_on:
            cur = torch.cuda.current_stream()
            ready = torch.cuda.Event()
            ready.record(cur)
            self._side.wait_event(ready)
            mod_ready = []
            with torch.cuda.stream(self._side):
                for k in range(D + 1):
                    name = f"blks.{k}.adaLN_modulation.1" if k < D else "out_layer.adaLN_modulation.1"
                    params_ready(1 + 4 * k, self._side)
                    w, b = self._lin(name)
                    g(a["s"], w, b, out=a["mod"][k] if k < D else a["fmod"], side=True)
                    ev = torch.cuda.Event()
                    ev.record(self._side)
                    mod_ready.append(ev)
            self._side_on = False
        # deferred split-K results (csrc/train_engine.hip, Ctx::Pend): the block input only with the side stream (else
        # the block's adaLN GEMM sits between producer and consumer and may use the same split-K scratch)
        def sr(on=True):
            return L.SplitResult() if (self.defer and on) else None
        w, b = self._lin("in_layer.0")
        pend = dict(split=sr(mod_ready is not None), bias=b)
        g(a["xp"], w, b, out=a["h"][0], defer=pend["split"])
        for k in range(D):
            p = f"blks.{k}."
            mod = a["mod"][k]
            if mod_ready is None:             # (else mod_ready[k] already orders this stream behind param_ready[1 + 4k])
                params_ready(1 + 4 * k)
                w, b = self._lin(p + "adaLN_modulation.1")
                g(a["s"], w, b, out=mod)
            else:
                torch.cuda.current_stream().wait_event(mod_ready[k])
            ops.ln_modulate(a["h"][k], mod[:, 0:H], mod[:, H:2 * H], out=a["xn1"][k], mean=a["mean1"][k], rstd=a["rstd1"][k],
                            **pend)
            params_ready(2 + 4 * k)           # (None unless the block is updated in pieces)
            w, b = self._lin(p + "attn.qkv")
            g(a["xn1"][k], w, b, out=a["qkv"][k])
            ops.attention_fwd(a["qkv"][k], a["cu"], a["C"], self.heads, out=a["o"][k], lse=a["lse"][k])
            w, b = self._lin(p + "attn.proj")
            mid = dict(split=sr(), bias=b, gate=mod[:, 2 * H:3 * H], res=a["h"][k], v_out=a["pa"][k])
            g(a["o"][k], w, b, out=a["hmid"][k], epi=L.EPI_GATE_RES, gate=mod[:, 2 * H:3 * H], res=a["h"][k], out2=a["pa"][k],
              defer=mid["split"])
            ops.ln_modulate(a["hmid"][k], mod[:, 3 * H:4 * H], mod[:, 4 * H:5 * H], out=a["xn2"][k], mean=a["mean2"][k],
                            rstd=a["rstd2"][k], **mid)
            params_ready(3 + 4 * k)
            w, b = self._lin(p + "mlp.fc1")
            g(a["xn2"][k], w, b, out=a["u"][k], epi=L.EPI_GELU_SAVE, out2=a["upre"][k])
            params_ready(4 + 4 * k)
            w, b = self._lin(p + "mlp.fc2")
            pend = dict(split=sr(mod_ready is not None), bias=b, gate=mod[:, 5 * H:6 * H], res=a["hmid"][k], v_out=a["pm"][k])
            g(a["u"][k], w, b, out=a["h"][k + 1], epi=L.EPI_GATE_RES, gate=mod[:, 5 * H:6 * H], res=a["hmid"][k], out2=a["pm"][k],
              defer=pend["split"])
        if mod_ready is None:
            params_ready(1 + 4 * D)
            w, b = self._lin("out_layer.adaLN_modulation.1")
            g(a["s"], w, b, out=a["fmod"])
        else:
            torch.cuda.current_stream().wait_event(mod_ready[D])
        ops.ln_modulate(a["h"][D], a["fmod"][:, 0:H], a["fmod"][:, H:2 * H], out=a["xnf"], mean=a["meanf"], rstd=a["rstdf"],
                        **pend)
        w, b = self._lin("out_layer.linear")
        g(a["xnf"], w, b, out=a["eps"])
        return a["eps"]

    # ------------------------------------------------------------------ backward
    def _dW(self, a, name, dY, X, rows=None, then=None):
        """grad[name.weight] = dY^T X ; grad[name.bias] = colsum(dY).  Both operands are read K-major ([batch, .]) as
        they lie: no transposed copies, and a batch that is not a multiple of the K-slab is zero-extended in the kernel."""
        if not self._side_on:
            self._dW_enqueue(a, name, dY, X, rows, False)
            if then is not None:
                then(False)
            return
        # Fork: the side stream starts when dY is final.  Join, one job late: from here on the caller's stream waits
        # for the PREVIOUS job.  Between two _dW calls nothing on the caller's stream rewrites the dY / X of the job
        # just issued (checked call by call in backward(); dmod alternates between two buffers for exactly that
        # reason), so one job may stay in flight while the dX chain goes on.
        cur = torch.cuda.current_stream()
        ready = torch.cuda.Event()
        ready.record(cur)
        self._side.wait_event(ready)
        with torch.cuda.stream(self._side):
            self._dW_enqueue(a, name, dY, X, rows, True)
            if then is not None:               # more work on the same inputs that nothing on the caller's chain needs
                then(True)
            done = torch.cuda.Event()
            done.record(self._side)
        if self._side_prev is not None:
            cur.wait_event(self._side_prev)
        self._side_prev = done

    def _dW_enqueue(self, a, name, dY, X, rows, side):
        gw = self.g[name + ".weight"]
        m = gw.shape[0] if rows is None else rows
        if m % 8 or gw.shape[1] % 8:         # odd gene counts: 16-byte column pieces do not exist, go through copies
            Cp = a["Cp"]
            bufY, bufX = a["tr"][1 if side else 0]
            dYT = ops.transpose(dY, out=bufY[:dY.shape[1]], rows_pad=Cp)
            XT = ops.transpose(X, out=bufX[:X.shape[1]], rows_pad=Cp)
            self._gemm(dYT[:m], XT[:gw.shape[1]], None, out=gw, side=side)
        else:
            self._gemm(dY, X[:, :gw.shape[1]] if X.shape[1] != gw.shape[1] else X, None, out=gw, M=m, a_kmajor=True,
                       w_kmajor=True, side=side)
        ops.colsum(dY[:, :m] if rows is not None else dY, self.g[name + ".bias"])

    def _group(self, problems, then=None, parts=None):
        """[(name, dY, X[, rows])]: every weight / bias gradient of the list in ONE launch (ops.dw_group), forked like
        _dW but not joined: backward() calls _side_join() where the group's buffers are first rewritten."""
        pr = []
        for item in problems:
            name, dY, X = item[:3]
            pr.append((dY, X, self.g[name + ".weight"], self.g[name + ".bias"], item[3] if len(item) > 3 else None))
        sumsq = None
        if parts is not None:                       # [buffer, floats used, start of the last slice]: this launch's slice
            n = ops.dw_group_parts(pr)
            sumsq = parts[0][parts[1]:parts[1] + n]
            assert sumsq.numel() == n
            parts[2], parts[1] = parts[1], parts[1] + n
        if not self._side_on:
            ops.dw_group(pr, sq_parts=sumsq)
            if then is not None:
                then(False)
            return
        cur = torch.cuda.current_stream()
        ready = torch.cuda.Event()
        ready.record(cur)
        self._side.wait_event(ready)
        with torch.cuda.stream(self._side):
            ops.dw_group(pr, sq_parts=sumsq)
            if then is not None:
                then(True)
            done = torch.cuda.Event()
            done.record(self._side)
        self._side_prev = done

    def _norm_parts(self):
        """The buffer the group launches leave their squared-norm parts in (stdiff_grad_ranges.parts)."""
        if self._parts is None:
            n = int(L.lib().stdiff_dit_backward_parts(self._c_weights()))
            assert n >= 0
            self._parts = torch.zeros(max(n, 4), dtype=torch.float32, device=self.device)
        return self._parts

    def _backward_grouped(self, a, after_bucket):
        """The launch sequence of backward_grouped() in csrc/train_engine.hip, issued from Python."""
        H, D, G = self.H, self.depth, self.G
        deps = a["deps"]
        dbrm, dbra = [a["dbr"], a["dbr_b"][0]], [a["dbr_b"][1], a["dbr_b"][2]]
        dupre, dqkv = [a["dupre"], a["du"]], [a["dqkv"], a["dqkv_b"]]
        odd = G % 8 != 0
        self._side_begin()
        if odd:
            self._dW(a, "out_layer.linear", deps, X=a["xnf"], rows=G)
        def sr():
            return L.SplitResult() if self.defer else None
        sp = sr()
        self._dX("out_layer.linear", deps, a["dxn"], defer=sp)
        if D > 0:
            ql = (D - 1) & 1
            ops.ln_modulate_bwd(a["dxn"], a["h"][D], a["meanf"], a["rstdf"], a["fmod"][:, H:2 * H], a["dh"],
                                a["dfmod"][:, 0:H], a["dfmod"][:, H:2 * H], accumulate=False,
                                gate=a["mod"][D - 1][:, 5 * H:6 * H], branch=a["pm"][D - 1], dbranch=dbrm[ql],
                                dgate=a["dmod"][ql][:, 5 * H:6 * H], split=sp)
        else:
            ops.ln_modulate_bwd(a["dxn"], a["h"][D], a["meanf"], a["rstdf"], a["fmod"][:, H:2 * H], a["dh"],
                                a["dfmod"][:, 0:H], a["dfmod"][:, H:2 * H], accumulate=False, split=sp)
        first = [("out_layer.adaLN_modulation.1", a["dfmod"], a["s"])]
        if not odd:
            first.append(("out_layer.linear", deps, a["xnf"], G))
        # the squared gradient norm: added by the group launches themselves (norm_in_group), else one pass per bucket
        in_group = self._norm is not None and self.norm_in_group and not odd
        gsq = [self._norm_parts(), 0, 0] if in_group else None

        def norm(i):
            if in_group:                               # the parts of the group launched last, on its stream
                ops.sum_parts(gsq[0][gsq[2]:gsq[1]], self._norm[0])
            elif self._norm is not None:
                lo, hi = self._norm[1][i]
                ops.grad_sumsq(self.flat_g[lo:hi], self._norm[0])

        def then_first(sd):
            norm(0)
            self._dX("out_layer.adaLN_modulation.1", a["dfmod"], a["ds32"], side=sd)
        self._group(first, then=then_first, parts=gsq)
        for k in range(D - 1, -1, -1):
            p, q = f"blks.{k}.", k & 1
            mod, dmod = a["mod"][k], a["dmod"][q]
            self._dX(p + "mlp.fc2", dbrm[q], dupre[q], epi=L.EPI_GELU_BWD, gate=a["upre"][k])
            sp = sr()
            self._dX(p + "mlp.fc1", dupre[q], a["dxn"], defer=sp)
            ops.ln_modulate_bwd(a["dxn"], a["hmid"][k], a["mean2"][k], a["rstd2"][k], mod[:, 4 * H:5 * H], a["dh"],
                                dmod[:, 3 * H:4 * H], dmod[:, 4 * H:5 * H], accumulate=True,
                                gate=mod[:, 2 * H:3 * H], branch=a["pa"][k], dbranch=dbra[q], dgate=dmod[:, 2 * H:3 * H],
                                split=sp)
            sp = sr()
            self._dX(p + "attn.proj", dbra[q], a["do"], defer=sp)
            ops.attention_bwd(a["qkv"][k], a["o"][k], a["do"], a["lse"][k], a["cu"], a["C"], self.heads,
                              dqkv=dqkv[q], delta=a["delta"], split=sp)
            sp = sr()
            self._dX(p + "attn.qkv", dqkv[q], a["dxn"], defer=sp)
            self._side_join()
            if after_bucket:
                after_bucket(D - 1 - k)
            if k > 0:
                ops.ln_modulate_bwd(a["dxn"], a["h"][k], a["mean1"][k], a["rstd1"][k], mod[:, H:2 * H], a["dh"],
                                    dmod[:, 0:H], dmod[:, H:2 * H], accumulate=True,
                                    gate=a["mod"][k - 1][:, 5 * H:6 * H], branch=a["pm"][k - 1], dbranch=dbrm[q ^ 1],
                                    dgate=a["dmod"][q ^ 1][:, 5 * H:6 * H], split=sp)
            else:
                ops.ln_modulate_bwd(a["dxn"], a["h"][k], a["mean1"][k], a["rstd1"][k], mod[:, H:2 * H], a["dh"],
                                    dmod[:, 0:H], dmod[:, H:2 * H], accumulate=True, split=sp)
            self._group([(p + "adaLN_modulation.1", dmod, a["s"]), (p + "mlp.fc1", dupre[q], a["xn2"][k]),
                         (p + "mlp.fc2", dbrm[q], a["u"][k]), (p + "attn.qkv", dqkv[q], a["xn1"][k]),
                         (p + "attn.proj", dbra[q], a["o"][k])],
                        then=lambda sd, p=p, dmod=dmod, k=k: (norm(D - k),
                                                              self._dX(p + "adaLN_modulation.1", dmod, a["ds32"],
                                                                       epi=L.EPI_ACCUM, side=sd),
                                                              # d loss / d h0 is final: its operand copy, beside the chain
                                                              k == 0 and ops.pack_rows(a["dh"], H, self.tdt, out=a["dhT_"])),
                        parts=gsq)
        self._side_join()
        if after_bucket:
            after_bucket(D)
        if D == 0:
            ops.pack_rows(a["dh"], H, self.tdt, out=a["dhT_"])
        ops.silu_bwd(a["ds32"], a["c"], a["dcT_"])        # straight into the operand format (no fp32 copy, no pack)
        self._dX("time_emb.mlp.2", a["dcT_"], a["de1"])
        ops.silu_bwd(a["de1"], a["a1"], a["da1"])
        self._side_on = False
        last = [("time_emb.mlp.2", a["dcT_"], a["e1"]), ("time_emb.mlp.0", a["da1"], a["e0"])]
        if not odd:
            last += [("in_layer.0", a["dhT_"], a["xp"]), ("cond_layer.0", a["dcT_"], a["yp"])]
        self._group(last, parts=gsq)
        if odd:
            self._dW_enqueue(a, "in_layer.0", a["dhT_"], a["xp"], None, False)
            self._dW_enqueue(a, "cond_layer.0", a["dcT_"], a["yp"], None, False)
        norm(D + 1)
        if after_bucket:
            after_bucket(D + 1)

    def _dX(self, name, dY, out, epi=L.EPI_BIAS, side=False, gate=None, defer=None):
        """out (+)= dY W  with W [out, in] read K-major (K = out features)."""
        return self._gemm(dY, self.pw.w[name + ".weight"], None, out=out, epi=epi, w_kmajor=True, side=side, gate=gate,
                          defer=defer)

    def backward(self, a, after_bucket=None, sumsq=None):
        """a['deps'] holds d loss / d eps as the packed operand [C, Gp]; fills self.flat_g (overwrite).
        after_bucket(i) is called when the i-th range of bucket_ranges() is final (DDP overlap).
        sumsq (fp32 [1], zeroed by the caller; grouped backward only): the squared gradient norm is accumulated bucket
        by bucket as the backward finishes them, beside the input-gradient chain, instead of in a pass of its own."""
        H, D = self.H, self.depth
        self._norm = None
        if sumsq is not None:
            assert "dbr_b" in a, "the folded gradient norm needs the grouped backward"
            self._norm = (sumsq, self.bucket_ranges())
        if self.c_enqueue:
            side = self._c_side()
            ca = self._c_acts(a)
            if self._norm is not None:
                gr = L.GradRanges()
                gr.sumsq = L.ptr(sumsq)
                for i, (lo, hi) in enumerate(self._norm[1]):
                    gr.lo[i], gr.count[i] = L.ptr(self.flat_g) + 4 * lo, hi - lo
                gr.in_group = int(os.environ.get("STDIFF_TRAIN_NORM_IN_GROUP", "1"))
                if gr.in_group and self.G % 8 == 0:
                    parts = self._norm_parts()
                    gr.parts, gr.parts_count = L.ptr(parts), parts.numel()
                a["_grad_ranges"] = gr
                ca.grad_ranges = L.C.pointer(gr)
            else:
                ca.grad_ranges = L.C.cast(None, L.C.POINTER(L.GradRanges))
            err = []

            def hook(i, _user):               # an exception must not unwind through the C frames
                try:
                    after_bucket(i)
                except BaseException as e:    # noqa: BLE001
                    err.append(e)
            cb = L.BucketFn(hook) if after_bucket else L.C.cast(None, L.BucketFn)
            L.check(L.lib().stdiff_dit_backward(self._c_weights(), self._c_acts(a), side, cb, None, L.stream()))
            if err:
                raise err[0]
            return
        if "dbr_b" in a:
            return self._backward_grouped(a, after_bucket)
        deps = a["deps"]
        self._side_begin()
        # ---- final layer
        self._dW(a, "out_layer.linear", deps, X=a["xnf"], rows=self.G)
        self._dX("out_layer.linear", deps, a["dxn"])
        ops.ln_modulate_bwd(a["dxn"], a["h"][D], a["meanf"], a["rstdf"], a["fmod"][:, H:2 * H], a["dh"],
                            a["dfmod"][:, 0:H], a["dfmod"][:, H:2 * H], accumulate=False)
        # d loss / d s accumulates over all adaLN projections and is needed only after the last block: its GEMMs ride
        # with the adaLN weight-gradient jobs (same dY), in issue order on whichever stream runs them
        self._dW(a, "out_layer.adaLN_modulation.1", a["dfmod"], X=a["s"],
                 then=lambda sd: self._dX("out_layer.adaLN_modulation.1", a["dfmod"], a["ds32"], side=sd))
        if after_bucket:
            self._side_join()
            after_bucket(0)
        for k in range(D - 1, -1, -1):
            p = f"blks.{k}."
            mod, dmod = a["mod"][k], a["dmod"][k & 1]
            # MLP branch: h_{k+1} = hmid + gate_mlp * pm
            ops.gate_bwd(a["dh"], mod[:, 5 * H:6 * H], a["pm"][k], a["dbr"], dmod[:, 5 * H:6 * H])
            self._dW(a, p + "mlp.fc2", a["dbr"], X=a["u"][k])
            self._dX(p + "mlp.fc2", a["dbr"], a["du"])
            ops.gelu_bwd(a["du"], a["upre"][k], a["dupre"])
            self._dW(a, p + "mlp.fc1", a["dupre"], X=a["xn2"][k])
            self._dX(p + "mlp.fc1", a["dupre"], a["dxn"])
            ops.ln_modulate_bwd(a["dxn"], a["hmid"][k], a["mean2"][k], a["rstd2"][k], mod[:, 4 * H:5 * H], a["dh"],
                                dmod[:, 3 * H:4 * H], dmod[:, 4 * H:5 * H], accumulate=True)
            # attention branch: hmid = h_k + gate_msa * pa
            ops.gate_bwd(a["dh"], mod[:, 2 * H:3 * H], a["pa"][k], a["dbr"], dmod[:, 2 * H:3 * H])
            self._dW(a, p + "attn.proj", a["dbr"], X=a["o"][k])
            self._dX(p + "attn.proj", a["dbr"], a["do"])
            ops.attention_bwd(a["qkv"][k], a["o"][k], a["do"], a["lse"][k], a["cu"], a["C"], self.heads,
                              dqkv=a["dqkv"], delta=a["delta"])
            self._dW(a, p + "attn.qkv", a["dqkv"], X=a["xn1"][k])
            self._dX(p + "attn.qkv", a["dqkv"], a["dxn"])
            ops.ln_modulate_bwd(a["dxn"], a["h"][k], a["mean1"][k], a["rstd1"][k], mod[:, H:2 * H], a["dh"],
                                dmod[:, 0:H], dmod[:, H:2 * H], accumulate=True)
            self._dW(a, p + "adaLN_modulation.1", dmod, X=a["s"],
                     then=lambda sd, p=p, dmod=dmod: self._dX(p + "adaLN_modulation.1", dmod, a["ds32"],
                                                              epi=L.EPI_ACCUM, side=sd))
            if after_bucket:
                self._side_join()
                after_bucket(D - k)
        # ---- embeddings: h0 = in_layer(x); c = time_emb(t) + cond_layer(y); s = silu(c)
        ops.pack_rows(a["dh"], H, self.tdt, out=a["dhT_"])
        self._dW(a, "in_layer.0", a["dhT_"], X=a["xp"])
        self._side_join()                      # ds32 is complete
        ops.silu_bwd(a["ds32"], a["c"], a["dc"])
        ops.pack_rows(a["dc"], H, self.tdt, out=a["dcT_"])
        self._dW(a, "cond_layer.0", a["dcT_"], X=a["yp"])
        self._dW(a, "time_emb.mlp.2", a["dcT_"], X=a["e1"])
        self._dX("time_emb.mlp.2", a["dcT_"], a["de1"])
        ops.silu_bwd(a["de1"], a["a1"], a["da1"])
        self._dW(a, "time_emb.mlp.0", a["da1"], X=a["e0"])
        self._side_join()                      # every gradient is final on the caller's stream from here on
        self._side_on = False
        if after_bucket:
            after_bucket(D + 1)


class TrainEngine(TrainCore):
    """One reference training iteration per step() (stDiff_train.py:59-81)."""

    def __init__(self, model, lr=1e-4, diffusion_step=1000, mask=None, weight_decay=0.0, betas=(0.9, 0.999),
                 eps=1e-8, max_norm=1.0, process_group=None, lazy_update=False):
        """lazy_update=True (single process): step() leaves its clip + AdamW update PENDING and the next step() issues
        it on a stream of its own, one range of the flat buffers per layer, while its forward is already running --
        the forward waits per layer for that layer's range (stdiff_adamw_range, stdiff_train_acts.param_ready).  Same
        arithmetic in the same order, so parameters, losses and gradients equal the eager sequence bit for bit; the
        parameters (and scal[3], the gradient norm) are current only after the next step() or finish()."""
        super().__init__(model)
        from .model.stDiff_scheduler import NoiseScheduler
        dev = self.device
        self.T = diffusion_step
        sched = NoiseScheduler(num_timesteps=diffusion_step, beta_schedule='cosine')      # stDiff_train.py:39-42
        self.sa = sched.sqrt_alphas_cumprod.to(dev)
        self.sb = sched.sqrt_one_minus_alphas_cumprod.to(dev)
        if mask is None:
            raise TypeError("mask is required (the reference crashes on torch.tensor(None), stDiff_train.py:69)")
        self.mask = torch.as_tensor(mask, dtype=torch.float32).to(dev).contiguous()
        self.lr, self.wd, self.betas, self.eps, self.max_norm = lr, weight_decay, betas, eps, max_norm
        self.m = torch.zeros_like(self.flat_p)
        self.v = torch.zeros_like(self.flat_p)
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        self.scal = torch.zeros(4, dtype=torch.float32, device=dev)
        # loss and squared gradient norm side by side: one fill launch zeroes both (they start at zero at the same point of
        # an iteration, between the forward and the loss)
        self._cells = torch.zeros(2, dtype=torch.float32, device=dev)
        self.loss, self.sumsq = self._cells[0:1], self._cells[1:2]
        self.pg = process_group
        self.dist = None
        self.world = 1
        # issue the per-bucket all-reduces even in a one-rank group (they are identities there): lets a single GPU
        # execute the stream-ordered RCCL bucket path (tests)
        self.force_buckets = os.environ.get("STDIFF_DDP_FORCE_BUCKETS") == "1"
        if process_group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()):
            import torch.distributed as dist
            self.dist = dist
            self.world = dist.get_world_size(process_group)
            # RCCL ("nccl") collectives are stream-ordered: all buckets are issued asynchronously and overlap the rest
            # of the backward pass.  gloo (tests / single-GPU rehearsal) stalls with several asynchronous CUDA
            # all-reduces in flight (observed: 8 x 12 MB) or with one issued behind a busy stream, so there the device
            # is synchronised and each bucket is reduced before the next is issued.
            self.async_buckets = dist.get_backend(process_group) == "nccl"
            if self.world > 1:
                # as torch DDP does at construction: every rank starts from rank 0's parameters (ranks that built the
                # model under different RNG states would otherwise train different models on averaged gradients)
                torch.cuda.synchronize(dev)
                dist.broadcast(self.flat_p, src=dist.get_global_rank(process_group, 0) if process_group is not None else 0,
                               group=process_group)
                if self.mirror is not None:
                    self.mirror.copy_(self.flat_p)
                self.pw.refresh()
        self.buckets = self.bucket_ranges()
        self._graphs, self._stream, self._stage_stream, self._flip = {}, None, None, 0
        self.fold_norm = os.environ.get("STDIFF_TRAIN_FOLD_NORM", "1") != "0"
        self.lazy = bool(lazy_update) and self.world == 1 and os.environ.get("STDIFF_TRAIN_LAZY", "1") != "0"
        self._pending, self._opt_stream = False, None
        # update ranges in the order the forward first touches them: embedding layers, block 0 .. depth-1, final layer
        # (the all-reduce buckets in reverse)
        per_block = [self.buckets[-1]] + self.buckets[-2:0:-1] + [self.buckets[0]]
        # The LAST blocks' ranges may be cut into four pieces in the order the forward uses them (adaLN projection -- last
        # in the flat buffer, first to be used --, qkv + proj, fc1, fc2), each with its own event: where the update sets
        # the pace (hidden 2048: 487 us per block range against 170 us of forward per block) the forward's tail behind the
        # last range shrinks from a whole block to one GEMM.  Every piece costs a launch and a cross-stream edge: measured
        # on the 512-cell iteration, pieces for the last 0 / 1 / 2 / 3 / 6 blocks give 6.04 / 6.01 / 6.00 / 6.13 / 6.48 ms at
        # hidden 2048 and 2.36 / 2.37 / 2.38 ms at hidden 1024 (where the forward, not the update, sets the pace) -- so the
        # last two blocks of wide models only (STDIFF_TRAIN_UPDATE_PIECES = number of blocks, 0 = none).
        e = os.environ.get("STDIFF_TRAIN_UPDATE_PIECES", "")
        self.piece_blocks = min(self.depth, int(e) if e else (2 if self.H >= 1536 else 0))
        self.update_ranges, self.update_slots = [per_block[0]], [0]        # (range, param_ready slot)
        for k in range(self.depth):
            lo, hi = per_block[1 + k]
            if k >= self.depth - self.piece_blocks:
                o = lambda n, k=k: self.offs[f"blks.{k}.{n}.weight"]
                assert lo == o("attn.qkv") < o("mlp.fc1") < o("mlp.fc2") < o("adaLN_modulation.1") < hi
                self.update_ranges += [(o("adaLN_modulation.1"), hi), (lo, o("mlp.fc1")), (o("mlp.fc1"), o("mlp.fc2")),
                                       (o("mlp.fc2"), o("adaLN_modulation.1"))]
                self.update_slots += [1 + 4 * k, 2 + 4 * k, 3 + 4 * k, 4 + 4 * k]
            else:
                self.update_ranges.append((lo, hi))
                self.update_slots.append(1 + 4 * k)
        self.update_ranges.append(per_block[-1])
        self.update_slots.append(1 + 4 * self.depth)
        lo0, hi0 = self.update_ranges[0]
        assert all(lo0 <= self.offs[n] < hi0 for n in self.pw.w if n not in self.mirrored), \
            "operands that need a repack after the optimizer must lie in the first update range"

    def step(self, x, x_cond, noise=None, timesteps=None, use_graph=False, join=True):
        """x, x_cond: fp32 [C, G] on the device.  Returns (loss[1], grad_norm[1]) device tensors (no host sync).

        use_graph=True replays the whole iteration (≈340 launches: add-noise .. AdamW .. operand repack) as ONE
        hipGraph per batch size; inputs are staged into fixed buffers first.  Single process only: with a process
        group the bucketed all-reduce is issued from the host between backward segments.

        join=False (graph replays only): the caller's stream is NOT made to wait for the iteration.  The inputs of the
        next step() are then staged (two copies, the noise and timestep draws) on a stream of their own while this
        iteration still runs, into the other of two staging sets, instead of between two replays on the replay stream
        (48 us of a 2.38 ms iteration).  The returned tensors -- and every engine buffer -- are then current on
        `self.stream` only: use accumulate() for running sums, and finish() / sync() before reading on another stream."""
        C = x.shape[0]
        dev = self.device
        self.pw.fused_stale = True       # the optimizer kernel rewrites the operand mirror without going through refresh()
        if use_graph and self.world == 1 and not (self.lazy and not self._pending):
            # (a lazy engine's first step has no pending update: enqueued eagerly, the graph is captured on the second)
            return self._step_graph(x, x_cond, noise, timesteps, join)
        a = self.acts(C)
        if noise is None:
            noise = torch.randn(x.shape, device=dev)                                   # stDiff_train.py:62
        if timesteps is None:
            timesteps = torch.randint(1, self.T, (C,), device=dev).long()              # :63  (t = 0 never drawn)
        self._enqueue(a, x, x_cond, noise, timesteps)
        return self.loss, self.scal[3:4]

    def _adamw_args(self):
        return ops.adamw_args(self.flat_p, self.flat_g, self.m, self.v, self.sumsq, self.step_dev, self.scal, self.lr,
                              max_norm=self.max_norm, grad_scale=1.0 / self.world, betas=self.betas, eps=self.eps,
                              weight_decay=self.wd, bf16_mirror=self.mirror)

    def _enqueue_pending_update(self):
        """The update the previous step() left pending, on the optimizer stream, one range per layer; returns the events
        recorded after each range (the forward's param_ready)."""
        if self._opt_stream is None:
            from .engine import aux_stream
            # (default priority: a high-priority optimizer stream made the iteration 40 % slower, and high-priority
            # chain streams hung a captured iteration on this stack -- neither is used)
            self._opt_stream = aux_stream(self.device, "opt")
        cur = torch.cuda.current_stream()
        start = torch.cuda.Event()
        start.record(cur)
        self._opt_stream.wait_event(start)
        events = [start]                             # kept alive with the others (a graph may be recording them)
        with torch.cuda.stream(self._opt_stream):
            args = self._adamw_args()
            ops.adamw_prepare(args)
            for i, (lo, hi) in enumerate(self.update_ranges):
                ops.adamw_range(args, lo, hi - lo)
                if i == 0:
                    self.refresh_weights()             # gene-padded in / cond operands: all inside the first range
                ev = torch.cuda.Event()
                ev.record(self._opt_stream)
                events.append(ev)
        self._fork_event = events.pop(0)
        self._last_update_event = events[-1]
        ready = [None] * (4 * self.depth + 2)              # stdiff_train_acts.param_ready: one slot per use point
        for slot, ev in zip(self.update_slots, events):
            ready[slot] = ev
        return ready

    def finish(self):
        """Applies the update a lazy step() left pending (no-op otherwise): call before reading parameters."""
        self.sync()                                 # iterations issued with join=False
        if self._pending:
            ops.clip_adamw(self.flat_p, self.flat_g, self.m, self.v, self.sumsq, self.step_dev, self.scal, self.lr,
                           max_norm=self.max_norm, grad_scale=1.0 / self.world, betas=self.betas, eps=self.eps,
                           weight_decay=self.wd, bf16_mirror=self.mirror)
            self.refresh_weights()
            self._pending = False

    def _enqueue(self, a, x, x_cond, noise, timesteps):
        ready = self._enqueue_pending_update() if (self.lazy and self._pending) else None
        ops.add_noise_splice(x, noise, timesteps, self.sa, self.sb, self.mask, xp=a["xp"])   # :65-70
        ops.pack_rows(x_cond, self.Gp, self.tdt, out=a["yp"])
        eps = self.forward(a, timesteps, param_ready=ready)                            # :72
        if ready is not None:
            a["_ready_events"] = (self._fork_event, ready)      # a captured graph keeps its events alive
            torch.cuda.current_stream().wait_event(self._last_update_event)      # the update has read every gradient
        hooked = self.world > 1 or (self.force_buckets and self.dist is not None)
        # single process + grouped backward: the gradient norm is accumulated inside the backward
        fold_norm = self.fold_norm and not hooked and "dbr_b" in a
        if fold_norm:
            self._cells.zero_()                  # loss and sumsq
        else:
            self.loss.zero_()
        ops.masked_mse(eps, noise, self.mask, self.loss, dp=a["deps"])                 # :75 (+ d loss / d eps)
        handles = []

        def reduce_bucket(i):
            lo, hi = self.buckets[i]
            if self.async_buckets:
                handles.append(self.dist.all_reduce(self.flat_g[lo:hi], group=self.pg, async_op=True))
            else:
                torch.cuda.synchronize(self.device)      # gloo + CUDA tensors behind a busy stream stalled on this stack
                self.dist.all_reduce(self.flat_g[lo:hi], group=self.pg)
        self.backward(a, after_bucket=reduce_bucket if hooked else None, sumsq=self.sumsq if fold_norm else None)       # :77
        for h in handles:
            h.wait()
        if not fold_norm:
            self.sumsq.zero_()
            ops.grad_sumsq(self.flat_g, self.sumsq)
        if self.lazy:
            self._pending = True                 # clip + AdamW ride under the next step's forward (or finish())
            return
        ops.clip_adamw(self.flat_p, self.flat_g, self.m, self.v, self.sumsq, self.step_dev, self.scal, self.lr,
                       max_norm=self.max_norm, grad_scale=1.0 / self.world, betas=self.betas, eps=self.eps,
                       weight_decay=self.wd, bf16_mirror=self.mirror)                  # :78-80
        self.refresh_weights()

    @property
    def stream(self):
        """The stream the captured iterations replay on (None before the first replay)."""
        return self._stream

    def sync(self):
        """Make the current stream wait for every iteration issued so far (needed after step(join=False))."""
        if self._stream is not None:
            torch.cuda.current_stream().wait_stream(self._stream)

    def accumulate(self, total, value):
        """total += value on the stream the iterations run on (the epoch's loss sum with step(join=False))."""
        if self._stream is None:
            total += value
            return
        with torch.cuda.stream(self._stream):
            total += value

    def _staging(self, C, k):
        sets = self._graphs.get(C)
        if sets is None:
            if len(self._graphs) > 3:
                self._graphs.clear()           # ops.Graph frees its exec handle when collected
            sets = self._graphs[C] = [None, None]
        if sets[k] is None:
            dev = self.device
            sets[k] = dict(
                x=torch.empty((C, self.G), device=dev), y=torch.empty((C, self.G), device=dev),
                noise=torch.empty((C, self.G), device=dev), t=torch.empty((C,), device=dev, dtype=torch.int64),
                graph=None, busy=None)
        return sets[k]

    def _stage(self, st, x, x_cond, noise, timesteps):
        st["x"].copy_(x)
        st["y"].copy_(x_cond)
        if noise is None:
            st["noise"].normal_()                                                  # stDiff_train.py:62
        else:
            st["noise"].copy_(noise)
        if timesteps is None:
            st["t"].random_(1, self.T)                                             # :63  (t = 0 never drawn)
        else:
            st["t"].copy_(timesteps)

    def _replay(self, st, C):
        if st["graph"] is None:
            a = st["a"] = self.acts(C)          # the graph records these addresses: keep the set alive with it
            g = ops.Graph()
            g.begin()
            try:
                self._enqueue(a, st["x"], st["y"], st["noise"], st["t"])
            finally:
                g.end()
            st["events"] = a.pop("_ready_events", None)        # (both staging sets share the activations)
            st["graph"] = g
        st["graph"].launch()

    def _step_graph(self, x, x_cond, noise, timesteps, join=True):
        C, dev = x.shape[0], self.device
        if self._stream is None:
            from .engine import lane_streams
            self._stream = lane_streams(dev, 1)[0]             # the legacy default stream cannot be captured
        cur = torch.cuda.current_stream()
        if join:
            st = self._staging(C, 0)
            self._stream.wait_stream(cur)
            with torch.cuda.stream(self._stream):
                self._stage(st, x, x_cond, noise, timesteps)
                self._replay(st, C)
                st["busy"] = None
            cur.wait_stream(self._stream)
            return self.loss, self.scal[3:4]
        # pipelined: staging set k alternates; its previous reader is the replay of two steps ago
        if self._stage_stream is None:
            from .engine import aux_stream
            self._stage_stream = aux_stream(dev, "stage")
        k = self._flip = self._flip ^ 1
        st = self._staging(C, k)
        S = self._stage_stream
        S.wait_stream(cur)                                     # the inputs are ready on the caller's stream
        if st["busy"] is not None:
            S.wait_event(st["busy"])
        with torch.cuda.stream(S):
            self._stage(st, x, x_cond, noise, timesteps)
        for tns in (x, x_cond, noise, timesteps):              # the caller may drop them before S has read them
            if tns is not None:
                tns.record_stream(S)
        self._stream.wait_stream(S)
        with torch.cuda.stream(self._stream):
            self._replay(st, C)
            st["busy"] = torch.cuda.Event()
            st["busy"].record(self._stream)
        return self.loss, self.scal[3:4]


class _DitFunction(torch.autograd.Function):
    """autograd bridge so that `loss(model(x, t, y)).backward()` works with the hand-written backward."""

    @staticmethod
    def forward(ctx, core, x, t, y, *params):
        C = x.shape[0]
        a = core.acts(C)
        if a.get("in_autograd"):
            # a forward of this batch size is still waiting for its backward: its saved activations (the packed
            # inputs xp / yp included: the in / cond weight gradients read them) must survive, so this call gets a
            # buffer set of its own (which replaces the cached one) BEFORE anything is written
            core._acts.pop(C, None)
            a = core.acts(C)
        a["in_autograd"] = True
        ops.pack_rows(x, core.Gp, core.tdt, out=a["xp"])
        ops.pack_rows(y, core.Gp, core.tdt, out=a["yp"])
        eps = core.forward(a, t)
        ctx.core, ctx.a = core, a
        return eps.clone()

    @staticmethod
    def backward(ctx, d_eps):
        core = ctx.core
        a = ctx.a                              # the buffer set of THIS forward (cache eviction cannot swap it)
        a["in_autograd"] = False
        ops.pack_rows(d_eps.contiguous().float(), core.Gp, core.tdt, out=a["deps"])
        core.backward(a)
        grads = tuple(core.g[n].clone() for n in core.names)
        return (None, None, None, None) + grads


def dit_forward_autograd(model, eng, x, t, y):
    core = getattr(model, "_train_core", None)
    if core is None or core.pw is not model.engine().pw:
        core = model._train_core = TrainCore(model)
    params = dict(model.named_parameters())
    return _DitFunction.apply(core, x, t, y, *[params[n] for n in core.names])
