"""DiT_stDiff on MI355X: the reference's module surface (constructor arguments, state-dict keys and shapes,
.forward(x, t, y, **kwargs)) over the hand-written gfx950 kernels of libstdiff_hip.so.

Reference: model/stDiff_model.py:145-243.  The nn.Module tree below exists only to own the parameters under the
reference's names (checkpoints `stDiff%d.pt` load with strict=True); none of its submodules is ever called --
forward() packs the parameters into MFMA operands and enqueues the fused kernels (stdiff_amd/engine.py)."""
import os

import torch
import torch.nn as nn

from .. import engine as E


class _Holder(nn.Module):
    """Parameter container; calling it is a bug (there is no eager path)."""

    def forward(self, *a, **k):
        raise RuntimeError("stdiff_amd modules are parameter holders; call DiT_stDiff.forward")


class _Act(_Holder):
    pass


class _Linear(nn.Linear):
    """nn.Linear as a parameter holder (same state-dict keys, same default init call sites as the reference's)."""

    def forward(self, *a, **k):       # a class-level override: a bound method stored ON the instance would be a
        raise RuntimeError("stdiff_amd modules are parameter holders; call DiT_stDiff.forward")   # reference cycle


def _linear(i, o):
    return _Linear(i, o, bias=True)


def _seq(*mods):
    s = nn.Sequential(*mods)
    return s


class _Attn(_Holder):
    def __init__(self, dim, num_heads):
        super().__init__()
        assert dim % num_heads == 0, 'dim should be divisible by num_heads'     # stDiff_model.py:13
        self.num_heads = num_heads
        self.qkv = _linear(dim, dim * 3)
        self.proj = _linear(dim, dim)


class _Mlp(_Holder):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = _linear(dim, hidden)
        self.fc2 = _linear(hidden, dim)


class _Block(_Holder):
    def __init__(self, dim, mlp_ratio, num_heads):
        super().__init__()
        self.attn = _Attn(dim, num_heads)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio))
        self.adaLN_modulation = _seq(_Act(), _linear(dim, 6 * dim))


class _Final(_Holder):
    def __init__(self, dim, out):
        super().__init__()
        self.linear = _linear(dim, out)
        self.adaLN_modulation = _seq(_Act(), _linear(dim, 2 * dim))


class _TimeEmb(_Holder):
    def __init__(self, dim, freq=256):
        super().__init__()
        self.mlp = _seq(_linear(freq, dim), _Act(), _linear(dim, dim))
        self.frequency_embedding_size = freq


class DiT_stDiff(nn.Module):
    """Denoiser: gene-space input/condition projections, sinusoidal time MLP, `depth` adaLN blocks with attention
    ACROSS THE CELLS of the batch, adaLN final projection (stDiff_model.py:145-243).

    Extra (not in the reference): ``compute_dtype`` -- "fp32" (default: exact-fp32 MFMA, matches the reference to
    ~1e-6), "bf16" (bf16 MFMA with fp32 accumulation) or, for sampling only, "fp16" (IEEE fp16 MFMA operands, fp32
    accumulation and residual stream: the same speed as bf16 with 8x smaller operand rounding); also settable through
    the STDIFF_DTYPE environment variable or ``model.set_compute_dtype()``."""

    def __init__(self, input_size, hidden_size, depth, dit_type, num_heads, classes, mlp_ratio=4.0, **kwargs):
        super().__init__()
        if dit_type != 'dit':
            raise KeyError(dit_type)                                  # BaseBlock lookup, stDiff_model.py:143,191
        if float(mlp_ratio) != 4.0:
            raise ValueError("the fused kernels are built for mlp_ratio=4.0 (the only value the reference uses)")
        self.input_size, self.hidden_size, self.depth = input_size, hidden_size, depth
        self.num_heads, self.classes, self.mlp_ratio, self.dit_type = num_heads, classes, mlp_ratio, dit_type
        self.in_layer = _seq(_linear(input_size, hidden_size))
        self.cond_layer = _seq(_linear(input_size, hidden_size))
        self.condi_emb = nn.Embedding(classes, hidden_size)           # unused by forward (stDiff_model.py:236)
        self.time_emb = _TimeEmb(hidden_size)
        self.blks = nn.ModuleList([_Block(hidden_size, mlp_ratio, num_heads) for _ in range(depth)])
        self.out_layer = _Final(hidden_size, input_size)
        self.compute_dtype = kwargs.get("compute_dtype", os.environ.get("STDIFF_DTYPE", "fp32"))
        self._engine = None
        self._engine_key = None
        self.initialize_weights()

    def initialize_weights(self):
        """Same distributions as stDiff_model.py:199-229: Xavier-uniform Linear weights with zero bias, N(0, 0.02)
        for the class table and the time MLP, zeros for every adaLN linear and the output linear."""
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
                nn.init.zeros_(m.bias)
        nn.init.normal_(self.condi_emb.weight, std=0.02)
        nn.init.normal_(self.time_emb.mlp[0].weight, std=0.02)
        nn.init.normal_(self.time_emb.mlp[2].weight, std=0.02)
        for lin in [b.adaLN_modulation[-1] for b in self.blks] + [self.out_layer.adaLN_modulation[-1],
                                                                   self.out_layer.linear]:
            nn.init.zeros_(lin.weight)
            nn.init.zeros_(lin.bias)

    # ------------------------------------------------------------------ engine plumbing
    def set_compute_dtype(self, name):
        E.parse_dtype(name)
        self.compute_dtype = name
        return self

    def _param_dict(self):
        return {k: v for k, v in self.named_parameters()}

    def engine(self):
        """DenoiserEngine over the current parameters; operand copies are refreshed when a parameter changed."""
        params = self._param_dict()
        dev = params["in_layer.0.weight"].device
        if dev.type != "cuda":
            raise RuntimeError("stdiff_amd runs on MI355X only: move the model with .to('cuda') (no CPU fallback)")
        dt = E.parse_dtype(self.compute_dtype)
        key = (dev, dt, tuple(p.data_ptr() for p in params.values()))
        ver = tuple(p._version for p in params.values())
        if self._engine is None or self._engine_key != key:
            pw = E.PackedWeights(params, self.input_size, self.hidden_size, self.depth, self.num_heads, dt)
            self._engine, self._engine_key, self._ver = E.DenoiserEngine(pw), key, ver
        elif self._ver != ver:
            self._engine.pw.refresh()
            self._ver = ver
        return self._engine

    def forward(self, x, t, y, **kwargs):
        """eps prediction for ONE batch of cells (attention runs across its rows).  x [C,G] any float dtype,
        t [C] integer timesteps, y [C,G] condition (stDiff_model.py:231-243; extra kwargs such as condi_flag are
        ignored exactly as the reference ignores them)."""
        if y is None:
            raise ValueError("DiT_stDiff needs a condition y (the reference crashes in cond_layer(None))")
        eng = self.engine()
        dev = eng.pw.device
        x = x.float().to(dev).contiguous()
        y = y.float().to(dev).contiguous()
        t = t.to(dev).long().contiguous()
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            from ..train_engine import dit_forward_autograd
            return dit_forward_autograd(self, eng, x, t, y)
        return eng.forward(x, t, y)
