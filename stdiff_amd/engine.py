"""Device-side engine behind DiT_stDiff / sample_stDiff: packed MFMA operands, workspaces, the enqueue of one
denoiser forward and the hipGraph-captured reverse-diffusion step.  All memory is torch-owned; all arithmetic is
in libstdiff_hip.so."""
import ctypes as C
import math

import os

import numpy as np
import torch

from . import _lib as L
from . import ops
from ._lib import BF16, F16, F32, c_dtype, check, lib_for, ptr, stream, torch_dtype

DTYPES = {"fp32": F32, "f32": F32, "float32": F32, "bf16": BF16, "bfloat16": BF16,
          "fp16": F16, "f16": F16, "float16": F16, "half": F16}


def parse_dtype(name):
    if isinstance(name, int):
        return name
    try:
        return DTYPES[str(name).lower()]
    except KeyError:
        raise ValueError(f"compute dtype must be 'fp32', 'bf16' or 'fp16' (sampling only), got {name!r}")


def sinusoid_freqs(device):
    """exp(-ln(1e4) * arange(128) / 128) with the reference's fp32 op sequence (stDiff_model.py:73-75)."""
    f = torch.exp(-math.log(10000) * torch.arange(start=0, end=128, dtype=torch.float32) / 128)
    return f.to(device)


class PackedWeights:
    """MFMA-operand copies of a DiT_stDiff state dict: weights in the compute dtype ([out, in] rows, the gene axis
    of in/cond layers zero-padded to a multiple of 64), biases fp32.  `refresh()` re-reads the parameters."""

    def __init__(self, params, G, H, depth, heads, dtype):
        if H % heads:
            raise AssertionError("dim should be divisible by num_heads")      # stDiff_model.py:13
        d = H // heads
        if H % 64 or d not in (32, 64, 128):
            raise ValueError(f"hidden_size {H} / num_heads {heads}: the gfx950 kernels need hidden % 64 == 0 and "
                             f"head_dim in (32, 64, 128)")
        self.params = params
        self.G, self.H, self.depth, self.heads, self.dtype = G, H, depth, heads, dtype
        self.Gp = ops.round_up(G, 64)
        self.tdt = torch_dtype(dtype)
        dev = params["in_layer.0.weight"].device
        self.device = dev
        self.w = {}
        for name, p in params.items():
            if name.endswith(".weight") and name != "condi_emb.weight":
                kp = self.Gp if name in ("in_layer.0.weight", "cond_layer.0.weight") else p.shape[1]
                self.w[name] = torch.empty((p.shape[0], kp), device=dev, dtype=self.tdt)
        self.freqs = sinusoid_freqs(dev)
        # fused-LN form of the adaLN projections (16-bit operands, STDIFF_FUSED_LN=0 switches it off): per block the
        # (shift, scale) rows of each LayerNorm in 256-row tiles [128 shift | 128 scale] and the two gate slices as one
        # [2H, H] matrix; made from the packed operands on first use and whenever they changed
        self.fused = None
        self.fused_stale = True
        if dtype != F32 and H % 128 == 0 and os.environ.get("STDIFF_FUSED_LN", "1") == "1":
            self.fused = {}
            for k in list(range(depth)) + ["f"]:
                for part in (("ln1", "ln2", "gate") if k != "f" else ("ln",)):
                    self.fused[f"{k}.{part}_w"] = torch.empty((2 * H, H), device=dev, dtype=self.tdt)
                    self.fused[f"{k}.{part}_b"] = torch.empty((2 * H,), device=dev, dtype=torch.float32)
            t = torch.arange(H // 128, device=dev).repeat_interleave(256) * 128
            j = torch.arange(256, device=dev).repeat(H // 128)
            self._tile_perm = t + (j % 128) + (j // 128) * H      # row of [shift(H) ; scale(H)] behind fused row i
        self.refresh()
        self._build_struct()

    def refresh(self, skip=()):
        for name, buf in self.w.items():
            if name not in skip:
                ops.pack_rows(self.params[name].detach(), buf.shape[1], buf.dtype, out=buf)
        self.fused_stale = True

    def ensure_fused(self):
        """Bring the fused-LN operand copies up to date with the packed adaLN operands (no-op when current)."""
        if self.fused is None or not self.fused_stale:
            return
        H, perm = self.H, self._tile_perm
        with torch.no_grad():
            for k in list(range(self.depth)) + ["f"]:
                name = f"blks.{k}.adaLN_modulation.1" if k != "f" else "out_layer.adaLN_modulation.1"
                W, b = self.w[name + ".weight"], self.b(name + ".bias")
                if k == "f":
                    torch.index_select(W, 0, perm, out=self.fused["f.ln_w"])
                    torch.index_select(b, 0, perm, out=self.fused["f.ln_b"])
                    continue
                for part, lo in (("ln1", 0), ("ln2", 3 * H)):
                    torch.index_select(W[lo:lo + 2 * H], 0, perm, out=self.fused[f"{k}.{part}_w"])
                    torch.index_select(b[lo:lo + 2 * H], 0, perm, out=self.fused[f"{k}.{part}_b"])
                self.fused[f"{k}.gate_w"][:H].copy_(W[2 * H:3 * H])
                self.fused[f"{k}.gate_w"][H:].copy_(W[5 * H:6 * H])
                self.fused[f"{k}.gate_b"][:H].copy_(b[2 * H:3 * H])
                self.fused[f"{k}.gate_b"][H:].copy_(b[5 * H:6 * H])
        self.fused_stale = False

    def adopt(self, views):
        """Replace operand buffers by caller-owned tensors of the same shape and dtype (the training engine's flat
        mirror, which its optimizer kernel keeps current); rebuilds the C structs that hold their addresses."""
        for name, t in views.items():
            assert t.shape == self.w[name].shape and t.dtype == self.w[name].dtype and t.is_contiguous()
            self.w[name] = t
        self.fused_stale = True
        self._build_struct()

    def b(self, name):
        return self.params[name].detach()

    def _build_struct(self):
        self.blocks = (L.BlockWeights * self.depth)()
        for k in range(self.depth):
            pre = f"blks.{k}."
            bw = self.blocks[k]
            for f, n in (("qkv", "attn.qkv"), ("proj", "attn.proj"), ("fc1", "mlp.fc1"), ("fc2", "mlp.fc2"),
                         ("ada", "adaLN_modulation.1")):
                setattr(bw, f + "_w", ptr(self.w[pre + n + ".weight"]))
                setattr(bw, f + "_b", ptr(self.b(pre + n + ".bias")))
            if self.fused is not None:
                for f in ("ln1", "ln2", "gate"):
                    setattr(bw, f + "_w", ptr(self.fused[f"{k}.{f}_w"]))
                    setattr(bw, f + "_b", ptr(self.fused[f"{k}.{f}_b"]))
        s = L.DitWeights()
        s.dtype, s.G, s.Gp, s.H, s.heads, s.depth = c_dtype(self.dtype), self.G, self.Gp, self.H, self.heads, self.depth
        s.in_w, s.in_b = ptr(self.w["in_layer.0.weight"]), ptr(self.b("in_layer.0.bias"))
        s.out_w, s.out_b = ptr(self.w["out_layer.linear.weight"]), ptr(self.b("out_layer.linear.bias"))
        s.fada_w = ptr(self.w["out_layer.adaLN_modulation.1.weight"])
        s.fada_b = ptr(self.b("out_layer.adaLN_modulation.1.bias"))
        s.blocks = C.cast(self.blocks, C.POINTER(L.BlockWeights))
        if self.fused is not None:
            s.lnf_w, s.lnf_b = ptr(self.fused["f.ln_w"]), ptr(self.fused["f.ln_b"])
        self.struct = s


class Workspace:
    """Activation buffers for R rows in chunks (chunk = one reference mini-batch; attention never crosses chunks).

    `share`: another Workspace with at least as many rows whose buffers this one aliases (row slabs of one sampler
    run one after the other on one stream, so they use the same memory; only the chunk offsets differ)."""

    BUFS = (("h", 1, True), ("s", 1, False), ("mod", 6, False), ("xn", 1, False), ("qkv", 3, False), ("o", 1, False),
            ("u", 4, False), ("fmod", 2, False))                 # name, width / H, fp32?

    def __init__(self, pw, chunk_lens, share=None):
        R = int(sum(chunk_lens))
        H, dev, tdt = pw.H, pw.device, pw.tdt
        if R > slab_rows_max(pw):
            raise ValueError(f"{R} rows in one workspace: the LDS-DMA GEMM addresses its operands with 32-bit byte "
                             f"offsets (use row slabs: engine.plan_slabs)")
        self.R, self.chunk_lens = R, list(chunk_lens)
        cu = np.concatenate([[0], np.cumsum(chunk_lens)]).astype(np.int32)
        self.cu_rows = torch.from_numpy(cu).to(dev)
        # fused-LN form (stdiff_hip.h, STDIFF_EPI_LN_MOD): the [R, 6H] modulation matrix is never materialised; row
        # statistics and the two gates of a block take its place.  The owner of shared buffers decides for its aliases.
        # Every workspace of a model takes the same form (the two agree to rounding, not to the bit, when H % 256 == 0).
        self.fused = share.fused if share is not None else pw.fused is not None
        bufs = [b for b in self.BUFS if not (self.fused and b[0] in ("mod", "fmod"))]
        if self.fused:
            bufs += [("gates", 2, False), ("ln_parts", 0, True)]
        self.mod = self.fmod = self.gates = self.ln_parts = None
        nparts = max(1, H // 256)            # partial row moments per row (stdiff_hip.h, stdiff_gemm_args)
        if share is not None:
            assert share.R >= R
            for name, width, f32 in bufs:
                setattr(self, name, getattr(share, name)[:R])
            self.eps = share.eps[:R]
        else:
            # ONE allocation for the whole set, carved at 2 MiB boundaries: a single large hipMalloc instead of nine
            specs = [(name, width * H if width else 2 * nparts, torch.float32 if f32 else tdt) for name, width, f32 in bufs]
            specs.append(("eps", pw.G, torch.float32))
            sizes = [ops.round_up(R * cols * torch.empty((), dtype=dt).element_size(), 2 << 20) for _, cols, dt in specs]
            self.arena = torch.empty(sum(sizes), device=dev, dtype=torch.uint8)
            off = 0
            for (name, cols, dt), nbytes in zip(specs, sizes):
                es = torch.empty((), dtype=dt).element_size()
                t = self.arena[off:off + R * cols * es].view(dt)
                setattr(self, name, t.view(R, cols))
                off += nbytes
        self._xp = None
        self._pw = pw
        s = L.DitWorkspace()
        s.R, s.n_chunks, s.max_chunk_rows = R, len(chunk_lens), int(max(chunk_lens)) if chunk_lens else 0
        s.cu_rows = ptr(self.cu_rows)
        for n in ("h", "mod", "xn", "qkv", "o", "u", "fmod", "ln_parts", "gates"):
            setattr(s, n, ptr(getattr(self, n)))
        self.struct = s

    @property
    def xp(self):
        """Packed [R, Gp] operand copy of the input (pad columns stay zero); made on first use -- the reverse sampler
        keeps its own full-length copy instead."""
        if self._xp is None:
            self._xp = torch.zeros((self.R, self._pw.Gp), device=self._pw.device, dtype=self._pw.tdt)
        return self._xp


def slab_rows_max(pw):
    """Most rows one launch may cover: the widest GEMM A operand ([R, 4H], the MLP hidden layer) must stay below
    2^32 bytes -- the LDS-DMA GEMM addresses operands through a buffer descriptor with 32-bit byte offsets (every
    other kernel, and every epilogue, uses 64-bit row offsets)."""
    return int(2.0e9 / (4 * pw.H) * (2 / (4 if pw.dtype == F32 else 2))) - 512


def slab_rows_limit(pw):
    """Rows per slab of the reverse sampler (STDIFF_SLAB_ROWS overrides the default), capped by slab_rows_max."""
    want = int(os.environ.get("STDIFF_SLAB_ROWS", "131072"))
    return max(1, min(slab_rows_max(pw), want))


def plan_slabs(chunk_lens, max_rows):
    """Cut a run of chunks into slabs of whole chunks with at most `max_rows` rows each (a chunk longer than that is a
    slab of its own): [(row_lo, row_hi, chunk_lens_of_the_slab)].  Slabs of a sampler are processed one after the other
    through ONE set of activation buffers: the working set stays a few GB however many spots a GPU holds."""
    out, cur, r0, r = [], [], 0, 0
    for c in chunk_lens:
        c = int(c)
        if cur and r - r0 + c > max_rows:
            out.append((r0, r, cur))
            cur, r0 = [], r
        cur.append(c)
        r += c
    if cur:
        out.append((r0, r, cur))
    return out


class DenoiserEngine:
    def __init__(self, pw):
        self.pw = pw
        self._ws = {}

    def workspace(self, chunk_lens, lane=0, share=None):
        """Cached activation buffers for this chunking; `lane` separates concurrent users of equal chunkings."""
        lens = tuple(int(c) for c in chunk_lens)
        key = (lens, int(lane), id(share) if share is not None else 0)
        ws = self._ws.get(key)
        if ws is None:
            if len(self._ws) > 8:
                self._ws.clear()
            ws = self._ws[key] = Workspace(self.pw, lens, share=share)
        return ws

    def slab_workspaces(self, chunk_lens, lane=0):
        """[(row_lo, row_hi, Workspace)] for the slabs of `chunk_lens` (plan_slabs): the largest slab owns the buffers,
        the others alias them."""
        slabs = plan_slabs(chunk_lens, slab_rows_limit(self.pw))
        if not slabs:
            return []
        big = max(slabs, key=lambda sl: sl[1] - sl[0])
        owner = self.workspace(big[2], lane)
        return [(lo, hi, owner if tuple(lens) == tuple(big[2]) else self.workspace(lens, lane, share=owner))
                for lo, hi, lens in slabs]

    # time_emb(t)  (TimestepEmbedder.forward, stDiff_model.py:82-85) -> fp32 [R,H]
    def time_embed(self, t):
        pw = self.pw
        e = ops.timestep_embedding(t, pw.freqs, out_dtype=pw.tdt)
        e = ops.gemm(e, pw.w["time_emb.mlp.0.weight"], pw.b("time_emb.mlp.0.bias"), epi=L.EPI_SILU)
        return ops.gemm(e, pw.w["time_emb.mlp.2.weight"], pw.b("time_emb.mlp.2.bias"), out_dtype=torch.float32)

    # cond_layer(y) (+ temb)  (stDiff_model.py:235-237) -> fp32 [R,H]
    def cond_embed(self, y, temb=None):
        pw = self.pw
        yp = ops.pack_rows(y, pw.Gp, pw.tdt)
        return ops.gemm(yp, pw.w["cond_layer.0.weight"], pw.b("cond_layer.0.bias"), out_dtype=torch.float32,
                        epi=L.EPI_ADD if temb is not None else L.EPI_BIAS, res=temb)

    def run(self, ws, xp=None, s=None):
        """eps = blocks/final(in_layer(xp); s)   (stDiff_model.py:240-243); xp / s default to the workspace's own."""
        xp = ws.xp if xp is None else xp
        s = ws.s if s is None else s
        assert xp.shape == (ws.R, self.pw.Gp) and xp.is_contiguous() and s.shape == (ws.R, self.pw.H)
        if ws.fused and self.pw.fused_stale:
            # the rebuild is a run of torch kernels: recorded into a hipGraph it would replay on every step and race
            # with the other lanes' graphs, so samplers bring the operands up to date BEFORE they capture
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("fused adaLN operands are stale inside a stream capture: call "
                                   "PackedWeights.ensure_fused() before capturing")
            self.pw.ensure_fused()
        check(lib_for(xp).stdiff_dit_forward(C.byref(self.pw.struct), C.byref(ws.struct), ptr(xp), ptr(s),
                                       ptr(ws.eps), ws.eps.stride(0), stream()))
        return ws.eps

    def forward(self, x, t, y, chunk_lens=None):
        """DiT_stDiff.forward(x, t, y) (stDiff_model.py:231-243) for inference; returns a fresh fp32 [R,G]."""
        R = x.shape[0]
        out = torch.empty((R, self.pw.G), device=self.pw.device, dtype=torch.float32)
        for lo, hi, ws in self.slab_workspaces(chunk_lens or [R]):
            ops.pack_rows(x[lo:hi], self.pw.Gp, self.pw.tdt, out=ws.xp)
            c = self.cond_embed(y[lo:hi], self.time_embed(t[lo:hi]))
            ops.silu_cond(c, out=ws.s)
            out[lo:hi].copy_(self.run(ws))
        return out


def step_coefficients(sched):
    """[T,5] fp32 (a, b, c1, c2, sigma) per timestep from the scheduler's own tables; sigma = variance**0.5 with
    get_variance's clip and t==0 -> 0 (stDiff_scheduler.py:92-100,134)."""
    var = sched.betas * (1.0 - sched.alphas_cumprod_prev) / (1.0 - sched.alphas_cumprod)
    var = var.clip(1e-20)
    sigma = var ** 0.5
    sigma[0] = 0.0
    return torch.stack([sched.sqrt_inv_alphas_cumprod, sched.sqrt_inv_alphas_cumprod_minus_one,
                        sched.posterior_mean_coef1, sched.posterior_mean_coef2, sigma], dim=1).contiguous().float()


class ReverseSampler:
    """The sample_stDiff loop (model/sample.py:56-95) on one device for `R` rows in reference chunks.

    Loop-invariant work is hoisted: cond_layer(x_cond) once (x_cond never changes, sample.py:9-16) and the time MLP
    once per timestep into a [T,H] table (all rows of a step share t, sample.py:11).  One reverse step
    (silu -> denoiser -> DDPM update + splice -> t -= 1) is captured into a hipGraph and replayed."""

    def __init__(self, engine, sched, chunk_lens, x_cond, gt, mask, pred_type="noise", row_offset=0, seed=0,
                 use_graph=True, lane=0, x_out=None):
        if pred_type not in ("noise", "x_start"):
            raise NotImplementedError()                                  # stDiff_scheduler.py:125-126
        pw = engine.pw
        self.engine, self.pw = engine, pw
        # row slabs (plan_slabs): the denoiser runs slab after slab through one set of activation buffers; what is
        # kept for every row is x, its packed operand copy xp, gt and the projected condition
        self.slabs = engine.slab_workspaces(chunk_lens, lane)
        self.R = int(sum(chunk_lens))
        self.T = len(sched)
        self.gt, self.mask = gt, mask
        self.pred_type = 0 if pred_type == "noise" else 1
        self.row_offset, self.seed, self.use_graph = int(row_offset), int(seed), use_graph
        dev = pw.device
        self.coef = step_coefficients(sched).to(dev)
        self.temb = engine.time_embed(torch.arange(self.T, device=dev, dtype=torch.int64))     # [T,H]
        self.ycond = torch.empty((self.R, pw.H), device=dev, dtype=torch.float32)              # [R,H]
        for lo, hi, _ in self.slabs:
            self.ycond[lo:hi].copy_(engine.cond_embed(x_cond[lo:hi]))
        self.xp = torch.zeros((self.R, pw.Gp), device=dev, dtype=pw.tdt)                       # pad columns stay zero
        self.x = x_out if x_out is not None else torch.empty((self.R, pw.G), device=dev, dtype=torch.float32)
        assert self.x.shape == (self.R, pw.G) and self.x.is_contiguous() and self.x.dtype == torch.float32
        self.t_dev = torch.zeros((1,), device=dev, dtype=torch.int32)
        self.graph = None
        self._graph_z = None
        self.stream = None

    def init(self, x_T=None):
        for lo, hi, _ in self.slabs:
            ops.init_sample(self.x[lo:hi], self.gt[lo:hi], self.mask, z=None if x_T is None else x_T[lo:hi],
                            seed=self.seed, row_offset=self.row_offset + lo, xp=self.xp[lo:hi])

    def _denoise_slab(self, lo, hi, ws):
        ops.silu_cond(self.ycond[lo:hi], temb=self.temb, t_index=self.t_dev, out=ws.s)
        return self.engine.run(ws, xp=self.xp[lo:hi])

    def _enqueue_step(self, t_first, z_base):
        for lo, hi, ws in self.slabs:
            eps = self._denoise_slab(lo, hi, ws)
            ops.ddpm_step(self.x[lo:hi], eps, self.gt[lo:hi], self.mask, self.coef, self.t_dev, t_first,
                          z_base=None if z_base is None else z_base[0, lo:hi], z_rows=self.R,
                          seed=self.seed, row_offset=self.row_offset + lo, xp=self.xp[lo:hi], pred_type=self.pred_type)
        ops.advance_timestep(self.t_dev)

    def denoise_only(self, t):
        """eps [R,G] for the current x at timestep t (teacher-forced parity checks)."""
        self.t_dev.fill_(int(t))
        out = torch.empty((self.R, self.pw.G), device=self.pw.device, dtype=torch.float32)
        for lo, hi, ws in self.slabs:
            out[lo:hi].copy_(self._denoise_slab(lo, hi, ws))
        return out

    def set_x(self, x):
        self.x.copy_(x)
        for lo, hi, _ in self.slabs:
            ops.pack_rows(self.x[lo:hi], self.pw.Gp, self.pw.tdt, out=self.xp[lo:hi])

    def run(self, timesteps, z=None):
        """timesteps: descending consecutive list (sample.py:57,63-64); z: optional injected noise [len-ish,R,G]
        indexed by (timesteps[0] - t)."""
        if not timesteps:
            return self.x
        t_first = int(timesteps[0])
        assert list(timesteps) == list(range(t_first, t_first - len(timesteps), -1))
        self.t_dev.fill_(t_first)
        self._t_first, self._z, self._t_next = t_first, z, t_first
        return self.run_more(len(timesteps))

    def run_more(self, n):
        """n further reverse steps from the device-side timestep counter."""
        t_first, z = self._t_first, self._z
        if n > self._t_next + 1:
            raise ValueError(f"{n} more steps requested but only {self._t_next + 1} timesteps remain")
        self._t_next -= n
        timesteps = range(n)
        if not self.use_graph:
            for _ in timesteps:
                self._enqueue_step(t_first, z)
            return self.x
        cur = torch.cuda.current_stream()
        self._prepare_graph(cur)
        with torch.cuda.stream(self.stream):
            for _ in timesteps:
                self.graph.launch()
        cur.wait_stream(self.stream)
        return self.x

    def _prepare_graph(self, cur):
        """Capture one reverse step on this sampler's own stream (the legacy default stream cannot be captured);
        afterwards that stream is ordered behind `cur`."""
        t_first, z = self._t_first, self._z
        if self.stream is None:
            self.stream = lane_streams(self.pw.device, 1)[0]
        self.pw.ensure_fused()               # eagerly, on the caller's stream, which every lane stream waits on below
        self.stream.wait_stream(cur)
        if self.graph is None or self._graph_z != (ptr(z), t_first):
            with torch.cuda.stream(self.stream):
                g = ops.Graph()
                g.begin()
                try:
                    self._enqueue_step(t_first, z)
                finally:
                    g.end()
            self.graph, self._graph_z = g, (ptr(z), t_first)


def concurrent_streams(device, n, candidates=12, spin_us=500):
    """`n` HIP streams (torch pool) whose hardware queues dispatch CONCURRENTLY.

    A process has few hardware queues (GPU_MAX_HW_QUEUES, 4 here) and the runtime maps streams onto them; two streams
    that share a queue -- or two queues the command processor serves from one pipe -- run their kernels one after the
    other, and the lanes then overlap nothing (measured: 6.2 instead of 5.2 ms per reverse step for the third and
    fourth stream of the pool).  Probe: 64 idle one-wave workgroups for `spin_us` on each of two streams finish in one
    period when the streams are concurrent and in two when they are not (stdiff_spin_probe)."""
    import time
    if n <= 1 or os.environ.get("STDIFF_LANE_PROBE", "1") == "0":
        return [torch.cuda.Stream(device=device) for _ in range(n)]

    def pair_ok(a, b):
        best = float("inf")
        for _ in range(2):
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            for st in (a, b):
                with torch.cuda.stream(st):
                    ops.spin_probe(64, spin_us)
            torch.cuda.synchronize(device)
            best = min(best, time.perf_counter() - t0)
        return best < 1.6e-6 * spin_us

    with torch.cuda.device(device):
        ops.spin_probe(1, 1)                             # loads the code object outside the timed launches
        chosen, spare = [], []
        for _ in range(candidates):
            st = torch.cuda.Stream(device=device)
            if all(pair_ok(st, c) for c in chosen):
                chosen.append(st)
                if len(chosen) == n:
                    return chosen
            else:
                spare.append(st)
    return (chosen + spare + [torch.cuda.Stream(device=device) for _ in range(n)])[:n]


# Streams of the samplers and trainers of this process, per device, made ONCE and shared by every object that needs a
# stream of its own (lane graphs, the capture stream of a training iteration, its side stream).
# Why they are not drawn afresh from torch's pool of 32 per object: a HIP stream that has carried a deep queue of graph
# replays keeps its pool of completion signals (KFD interrupt events) for the life of the process, and once about 21
# DIFFERENT streams of a process had done so, no new GPU process could start on that GPU: its hsa_init failed to create
# its first event and ROCr's async-events thread dereferenced the null HsaEvent (SIGSEGV; scripts/diag/, DESIGN.md 6a').
# A 5-fold run through both 16-bit formats made 30 such streams.  With the cache a process uses 2 (lanes) + 1 (side).
_STREAMS = {}


def _dev_index(device):
    d = torch.device(device)
    return d.index if d.index is not None else torch.cuda.current_device()


def lane_streams(device, n):
    """The `n` lane streams of `device` (concurrently dispatching, see concurrent_streams), the same ones at every call."""
    key = (_dev_index(device), "lanes")
    have = _STREAMS.get(key, [])
    if len(have) < n:
        have = _STREAMS[key] = concurrent_streams(device, max(n, 2))      # 2 from the start: the usual lane count
    return have[:n]


def aux_stream(device, name):
    """A cached non-default stream of `device` for role `name` ('side': the weight-gradient branch of a captured
    training iteration; 'opt': the optimizer update that runs under the next forward)."""
    key = (_dev_index(device), name)
    st = _STREAMS.get(key)
    if st is None:
        st = _STREAMS[key] = torch.cuda.Stream(device=device)
    return st


def default_lanes(chunk_lens):
    """Concurrent lanes per device: 2 when each lane still fills the chip (STDIFF_LANES overrides)."""
    env = os.environ.get("STDIFF_LANES")
    if env:
        return max(1, min(int(env), len(chunk_lens)))
    return 2 if len(chunk_lens) >= 8 and sum(chunk_lens) >= 8192 else 1


class LanedSampler:
    """ReverseSampler over `lanes` contiguous groups of chunks, each on its own HIP stream with its own hipGraph.

    Chunks are independent in the reverse loop (sample.py:9-16: the denoiser sees one mini-batch at a time), so the
    lanes never synchronise.  Two lanes keep the matrix cores busy while the other lane's GEMM epilogue pushes its C
    tile through the write path, fill the tail of a kernel whose tile count does not divide the CU count, and run
    the HBM-bound LayerNorm / DDPM-step kernels of one lane under the GEMMs of the other.  Noise is keyed by the
    GLOBAL row, so the result does not depend on the lane count (tests/test_gpu_model.py)."""

    def __init__(self, engine, sched, chunk_lens, x_cond, gt, mask, pred_type="noise", row_offset=0, seed=0,
                 use_graph=True, lanes=None):
        chunk_lens = [int(c) for c in chunk_lens]
        lanes = default_lanes(chunk_lens) if lanes is None else max(1, min(int(lanes), max(1, len(chunk_lens))))
        R = sum(chunk_lens)
        self.x = torch.empty((R, engine.pw.G), device=engine.pw.device, dtype=torch.float32)
        self.parts = []
        per, extra, c0, r0 = len(chunk_lens) // lanes, len(chunk_lens) % lanes, 0, 0
        counts = [per + (1 if i < extra else 0) for i in range(lanes)]
        split = os.environ.get("STDIFF_LANE_SPLIT")          # experiments: chunks per lane, e.g. "16,16,8"
        if split:
            counts = [int(v) for v in split.split(",")]
            if sum(counts) != len(chunk_lens) or min(counts) < 1:
                raise ValueError(f"STDIFF_LANE_SPLIT={split!r} does not partition {len(chunk_lens)} chunks")
            lanes = len(counts)
        for i in range(lanes):
            n = counts[i]
            lens = chunk_lens[c0:c0 + n]
            r1 = r0 + sum(lens)
            self.parts.append(ReverseSampler(engine, sched, lens, x_cond[r0:r1], gt[r0:r1], mask, pred_type=pred_type,
                                             row_offset=row_offset + r0, seed=seed, use_graph=use_graph, lane=i,
                                             x_out=self.x[r0:r1]))
            c0, r0 = c0 + n, r1
        self.bounds = [(p.row_offset - row_offset, p.row_offset - row_offset + p.R) for p in self.parts]
        self.use_graph = use_graph
        if use_graph and len(self.parts) > 1:           # lane streams that really overlap (see concurrent_streams)
            for p, st in zip(self.parts, lane_streams(engine.pw.device, len(self.parts))):
                p.stream = st

    def init(self, x_T=None):
        for p, (a, b) in zip(self.parts, self.bounds):
            p.init(None if x_T is None else x_T[a:b])

    def run(self, timesteps, z=None):
        if z is not None and len(self.parts) > 1:
            raise ValueError("injected step noise needs lanes=1 (its layout is [step, all rows, G])")
        if not timesteps:
            return self.x
        t_first = int(timesteps[0])
        assert list(timesteps) == list(range(t_first, t_first - len(timesteps), -1))
        for p in self.parts:
            p.t_dev.fill_(t_first)
            p._t_first, p._z, p._t_next = t_first, z, t_first
        return self.run_more(len(timesteps))

    def run_more(self, n):
        if len(self.parts) == 1 or not self.use_graph:
            for p in self.parts:
                p.run_more(n)
            return self.x
        cur = torch.cuda.current_stream()
        for p in self.parts:
            if n > p._t_next + 1:
                raise ValueError(f"{n} more steps requested but only {p._t_next + 1} timesteps remain")
            p._t_next -= n
            p._prepare_graph(cur)
        for _ in range(n):                      # interleave the launches so the lanes advance together
            for p in self.parts:
                with torch.cuda.stream(p.stream):
                    p.graph.launch()
        for p in self.parts:
            cur.wait_stream(p.stream)
        return self.x
