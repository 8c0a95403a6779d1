"""Tensor-level wrappers over the C ABI (one function per entry point of include/stdiff_hip.h).
Tensors must live on the GPU; outputs are allocated with torch when not supplied."""
import ctypes as C

import torch

from . import _lib as L
from ._lib import F32, check, dt_of, lib, lib_for, ptr, require_gpu, stream


def round_up(a, b):
    return (a + b - 1) // b * b


def gemm(A, W, bias=None, out=None, out_dtype=None, epi=L.EPI_BIAS, gate=None, res=None, out2=None, M=None,
         workspace=None, a_kmajor=False, w_kmajor=False, static_scratch=False, ln_parts=None, stats_out=None,
         defer=None):
    """out[M,N] = epi(A[M,K] @ W[N,K]^T + bias). A/W: same dtype (fp32 or bf16), K-contiguous rows -- or, with
    a_kmajor / w_kmajor, given as [K, M] / [K, N] (K rows; need not be a multiple of the K-slab).
    static_scratch=True: stdiff_gemm_static_scratch -- split-K partials in the library's own device array, nothing is
    allocated by the library (`out` must be given for the CALLER not to allocate either).
    defer: an L.SplitResult; when the launch splits K (bias / gated-residual epilogues only) the reduce launch is skipped
    and `defer` describes the partial products for a *_split consumer (defer.slices == 0: `out` is complete)."""
    require_gpu(A, W, bias, out, gate, res, out2, workspace)
    bk = 64 if A.element_size() == 2 else 16
    kA = A.shape[0] if a_kmajor else A.shape[1]
    kW = W.shape[0] if w_kmajor else W.shape[1]
    if a_kmajor and w_kmajor:
        assert kA == kW
        k_valid, K = kA, round_up(kA, bk)
    elif a_kmajor or w_kmajor:
        k_valid = kA if a_kmajor else kW            # the row-major operand carries the (zero-)padded K
        K = kW if a_kmajor else kA
        assert k_valid <= K, (k_valid, K)
    else:
        assert kA == kW
        k_valid = K = kA
    M = (A.shape[1] if a_kmajor else A.shape[0]) if M is None else M
    N = W.shape[1] if w_kmajor else W.shape[0]
    if out is None:
        out = torch.empty((M, N // 2 if epi == L.EPI_LN_MOD else N), device=A.device, dtype=out_dtype or A.dtype)
    a = L.GemmArgs()
    a.M, a.N, a.K = M, N, K
    a.a_layout, a.w_layout, a.k_valid = int(a_kmajor), int(w_kmajor), k_valid
    a.A, a.lda, a.a_dtype = ptr(A), A.stride(0), dt_of(A)
    a.W, a.ldw = ptr(W), W.stride(0)
    a.bias = ptr(bias)
    a.C, a.ldc, a.c_dtype = ptr(out), out.stride(0), dt_of(out)
    a.epi = epi
    if gate is not None:
        a.gate, a.ldgate = ptr(gate), gate.stride(0)
    if res is not None:
        a.res, a.ldres = ptr(res), res.stride(0)
    if out2 is not None:
        a.C2, a.ldc2 = ptr(out2), out2.stride(0)
    if ln_parts is not None:           # EPI_LN_MOD: partial row moments of `res`, fp32 [M, parts, 2] (row_stats: 1 part)
        require_gpu(ln_parts)
        assert ln_parts.dtype == torch.float32 and ln_parts.is_contiguous() and ln_parts.shape[0] >= M
        a.ln_parts, a.ln_nparts = ptr(ln_parts), ln_parts.shape[1]
    if stats_out is not None:          # gated-residual epilogue also writes the partial row moments of its output
        require_gpu(stats_out)
        assert stats_out.dtype == torch.float32 and stats_out.is_contiguous() and stats_out.shape == (M, N // 256, 2)
        a.stats_out = ptr(stats_out)
    if workspace is not None:          # fp32 scratch for split-K (small outputs with a long K)
        a.workspace, a.workspace_bytes = ptr(workspace), workspace.numel() * workspace.element_size()
    if defer is not None:
        a.defer = C.pointer(defer)
    lb = lib_for(A, W, out, gate, out2)
    check((lb.stdiff_gemm_static_scratch if static_scratch else lb.stdiff_gemm)(C.byref(a), stream()))
    return out


def _dw_group_args(problems, C_rows):
    a = L.DwGroupArgs()
    assert 1 <= len(problems) <= 8
    first = problems[0][0]
    a.n, a.C, a.dtype = len(problems), first.shape[0] if C_rows is None else C_rows, dt_of(first)
    for i, pr in enumerate(problems):
        dY, X, gw, gb = pr[:4]
        require_gpu(dY, X, gw, gb)
        assert dY.dtype == first.dtype and X.dtype == first.dtype and gw.dtype == torch.float32 and gw.is_contiguous()
        assert dY.shape[0] >= a.C and X.shape[0] >= a.C and dY.stride(1) == 1 and X.stride(1) == 1
        q = a.p[i]
        q.M, q.N = (gw.shape[0] if len(pr) < 5 or pr[4] is None else pr[4]), gw.shape[1]
        assert dY.shape[1] >= q.M and X.shape[1] >= q.N
        q.dY, q.lddy, q.X, q.ldx = ptr(dY), dY.stride(0), ptr(X), X.stride(0)
        q.gw, q.ldgw, q.gb = ptr(gw), gw.stride(0), ptr(gb)
    return a, first


def dw_group_parts(problems, C_rows=None):
    """Number of floats dw_group(problems, sq_parts=...) writes."""
    a, first = _dw_group_args(problems, C_rows)
    n = lib_for(first).stdiff_dw_group_parts(C.byref(a))
    if n < 0:
        raise ValueError("stdiff_dw_group_parts: bad problem list")
    return int(n)


def dw_group(problems, C_rows=None, sq_parts=None):
    """One launch for several weight gradients over the same batch: for each (dY, X, gw, gb[, M]) in `problems`
    gw[M, N] = dY[:, :M]^T X[:, :N] and gb[M] = colsum(dY[:, :M]) (gb may be None).  dY [C, >= M], X [C, >= N] of the
    operand dtype, gw / gb fp32, overwritten.  sq_parts (fp32 [>= dw_group_parts(problems)], optional): overwritten with
    per-wave sums of squares of everything written -- their sum is the launch's squared gradient norm (sum_parts)."""
    a, first = _dw_group_args(problems, C_rows)
    if sq_parts is not None:
        require_gpu(sq_parts)
        n = lib_for(first).stdiff_dw_group_parts(C.byref(a))
        assert sq_parts.dtype == torch.float32 and sq_parts.is_contiguous() and 0 <= n <= sq_parts.numel()
        a.sq_parts = ptr(sq_parts)
    check(lib_for(first).stdiff_dw_group(C.byref(a), stream()))


def sum_parts(parts, out, accumulate=True):
    """out[0] (+)= parts.sum(), one workgroup, fixed order (the same bits every run)."""
    require_gpu(parts, out)
    assert parts.dtype == torch.float32 and out.dtype == torch.float32 and parts.is_contiguous()
    check(lib().stdiff_sum_parts(ptr(parts), parts.numel(), ptr(out), int(accumulate), stream()))


def transpose(x, out=None, rows_pad=None):
    """out[c, r] = x[r, c]; columns rows..rows_pad-1 of out are zero-filled (K padding of the backward GEMMs)."""
    require_gpu(x, out)
    r, c = x.shape
    rows_pad = r if rows_pad is None else rows_pad
    if out is None:
        out = torch.empty((c, rows_pad), device=x.device, dtype=x.dtype)
    check(lib_for(x, out).stdiff_transpose(ptr(x), x.stride(0), ptr(out), out.stride(0), r, c, rows_pad, dt_of(x), stream()))
    return out


def attention_fwd(qkv, cu_rows, max_chunk, heads, out=None, lse=None):
    """qkv [R,3H] -> out [R,H]; cu_rows int32 device [n_chunks+1]."""
    require_gpu(qkv, cu_rows, out, lse)
    R, H3 = qkv.shape
    H = H3 // 3
    if out is None:
        out = torch.empty((R, H), device=qkv.device, dtype=qkv.dtype)
    check(lib_for(qkv, out).stdiff_attention_fwd(ptr(qkv), qkv.stride(0), ptr(out), out.stride(0), ptr(lse), ptr(cu_rows),
                                     cu_rows.numel() - 1, max_chunk, heads, H // heads, dt_of(qkv), stream()))
    return out


def ln_modulate(x, shift, scale, out=None, out_dtype=None, mean=None, rstd=None, split=None, bias=None, gate=None,
                res=None, v_out=None):
    """LayerNorm(x, eps=1e-6, no affine) * (1 + scale) + shift; shift/scale are column views of the adaLN output.
    split (an L.SplitResult with slices > 1): x is not read but FORMED first -- v = sum of the slices + bias,
    x = res + gate * v (x = v without a gate), v_out = v -- and written (stdiff_ln_modulate_split)."""
    require_gpu(x, shift, scale, out, bias, gate, res, v_out)
    R, H = x.shape
    assert shift.stride(0) == scale.stride(0) and shift.dtype == scale.dtype
    if out is None:
        out = torch.empty((R, H), device=x.device, dtype=out_dtype or shift.dtype)
    if split is not None and split.slices > 1:
        check(lib_for(shift, out).stdiff_ln_modulate_split(
            C.byref(split), ptr(bias), ptr(gate), gate.stride(0) if gate is not None else 0, ptr(res),
            res.stride(0) if res is not None else 0, ptr(x), x.stride(0), ptr(v_out),
            v_out.stride(0) if v_out is not None else 0, ptr(shift), ptr(scale), shift.stride(0), dt_of(shift), ptr(out),
            out.stride(0), dt_of(out), ptr(mean), ptr(rstd), R, H, stream()))
        return out
    check(lib_for(shift, out).stdiff_ln_modulate(ptr(x), x.stride(0), ptr(shift), ptr(scale), shift.stride(0), dt_of(shift),
                                   ptr(out), out.stride(0), dt_of(out), ptr(mean), ptr(rstd), R, H, stream()))
    return out


def row_stats(x, parts=None):
    """Row moments of x fp32 [R, H] as ONE part per row: fp32 [R, 1, 2] = (mean, sum of squared deviations) -- the
    statistics of LayerNorm(eps 1e-6, biased variance) in the form the LN_MOD GEMM epilogue reads."""
    require_gpu(x, parts)
    R, H = x.shape
    if parts is None:
        parts = torch.empty((R, 1, 2), device=x.device, dtype=torch.float32)
    assert parts.dtype == torch.float32 and parts.is_contiguous() and parts.numel() >= 2 * R
    check(lib().stdiff_row_stats(ptr(x), x.stride(0), ptr(parts), R, H, stream()))
    return parts


def silu_cond(ycond, temb=None, t_index=None, out=None, out_dtype=torch.float32):
    require_gpu(ycond, temb, t_index, out)
    R, H = ycond.shape
    if out is None:
        out = torch.empty((R, H), device=ycond.device, dtype=out_dtype)
    check(lib_for(out).stdiff_silu_cond(ptr(temb), ptr(t_index), ptr(ycond), ycond.stride(0), ptr(out), out.stride(0),
                                 dt_of(out), R, H, stream()))
    return out


def timestep_embedding(t, freqs, out=None, out_dtype=torch.float32):
    require_gpu(t, freqs, out)
    assert t.dtype == torch.int64
    R = t.numel()
    if out is None:
        out = torch.empty((R, 256), device=t.device, dtype=out_dtype)
    check(lib_for(out).stdiff_timestep_embedding(ptr(t), ptr(freqs), ptr(out), out.stride(0), dt_of(out), R, stream()))
    return out


def pack_rows(src, cpad, dtype, out=None):
    require_gpu(src, out)
    R, Cc = src.shape
    if out is None:
        out = torch.empty((R, cpad), device=src.device, dtype=dtype)
    check(lib_for(out).stdiff_pack_rows(ptr(src), src.stride(0), ptr(out), out.stride(0), dt_of(out), R, Cc, cpad, stream()))
    return out


def gather_rows2(a, b, idx, out_a=None, out_b=None):
    """(a[idx], b[idx]) for fp32 [N, C] matrices on the device and an int64 device index vector (b may be None)."""
    require_gpu(a, b, idx, out_a, out_b)
    R, Cc = idx.shape[0], a.shape[1]
    assert idx.dtype == torch.int64 and a.dtype == torch.float32 and (b is None or (b.dtype == a.dtype and b.shape == a.shape))
    if out_a is None:
        out_a = torch.empty((R, Cc), device=a.device, dtype=torch.float32)
    if b is not None and out_b is None:
        out_b = torch.empty((R, Cc), device=a.device, dtype=torch.float32)
    check(lib().stdiff_gather_rows2(ptr(a), a.stride(0), ptr(b), b.stride(0) if b is not None else 0, ptr(idx),
                                    ptr(out_a), out_a.stride(0), ptr(out_b), out_b.stride(0) if b is not None else 0,
                                    R, Cc, stream()))
    return out_a, out_b


def ddpm_step(x, eps, gt, mask, coef, t_dev, t_first, z_base=None, seed=0, row_offset=0, xp=None, pred_type=0, z_rows=0):
    require_gpu(x, eps, gt, mask, coef, t_dev, z_base, xp)
    a = L.StepArgs()
    a.x, a.eps, a.gt, a.mask, a.coef = ptr(x), ptr(eps), ptr(gt), ptr(mask), ptr(coef)
    a.t_dev, a.t_first = ptr(t_dev), t_first
    a.z_base, a.seed, a.row_offset = ptr(z_base), seed, row_offset
    if xp is not None:
        a.xp, a.ldxp, a.xp_dtype = ptr(xp), xp.stride(0), dt_of(xp)
    a.R, a.G, a.pred_type, a.z_rows = x.shape[0], x.shape[1], pred_type, int(z_rows)
    check(lib_for(xp).stdiff_ddpm_step(C.byref(a), stream()))


def advance_timestep(t_dev):
    check(lib().stdiff_advance_timestep(ptr(t_dev), stream()))


def spin_probe(blocks, microseconds):
    """`blocks` idle one-wave workgroups for `microseconds` on the current stream (stream-concurrency probe)."""
    check(lib().stdiff_spin_probe(int(blocks), int(microseconds), stream()))


def init_sample(x, gt, mask, z=None, seed=0, row_offset=0, xp=None):
    require_gpu(x, gt, mask, z, xp)
    check(lib_for(xp).stdiff_init_sample(ptr(x), ptr(z), seed, row_offset, ptr(gt), ptr(mask), ptr(xp),
                                   xp.stride(0) if xp is not None else 0, dt_of(xp) if xp is not None else F32,
                                   x.shape[0], x.shape[1], stream()))


def add_noise_splice(x, noise, t, sqrt_ac, sqrt_1mac, mask, x_noisy=None, xp=None):
    require_gpu(x, noise, t, sqrt_ac, sqrt_1mac, mask, x_noisy, xp)
    check(lib_for(xp).stdiff_add_noise_splice(ptr(x), ptr(noise), ptr(t), ptr(sqrt_ac), ptr(sqrt_1mac), ptr(mask),
                                        ptr(x_noisy), ptr(xp), xp.stride(0) if xp is not None else 0,
                                        dt_of(xp) if xp is not None else F32, x.shape[0], x.shape[1], stream()))


def masked_mse(pred, noise, mask, loss, dpred=None, dp=None, grad_scale=1.0):
    require_gpu(pred, noise, mask, loss, dpred, dp)
    check(lib_for(dp).stdiff_masked_mse(ptr(pred), ptr(noise), ptr(mask), ptr(loss), ptr(dpred), ptr(dp),
                                  dp.stride(0) if dp is not None else 0, dt_of(dp) if dp is not None else F32,
                                  grad_scale, pred.shape[0], pred.shape[1], stream()))


def attention_bwd(qkv, out, dout, lse, cu_rows, max_chunk, heads, dqkv=None, delta=None, split=None):
    """split (an L.SplitResult with slices > 1): dout is first FORMED from the partial products (and written)."""
    require_gpu(qkv, out, dout, lse, cu_rows, dqkv, delta)
    R, H3 = qkv.shape
    H = H3 // 3
    if dqkv is None:
        dqkv = torch.empty_like(qkv)
    if delta is None:
        delta = torch.empty((R, heads), device=qkv.device, dtype=torch.float32)
    if split is not None and split.slices > 1:
        check(lib_for(qkv, out, dout, dqkv).stdiff_attention_bwd_split(
            ptr(qkv), qkv.stride(0), ptr(out), out.stride(0), C.byref(split), ptr(dout), dout.stride(0), ptr(lse),
            ptr(delta), ptr(dqkv), dqkv.stride(0), ptr(cu_rows), cu_rows.numel() - 1, max_chunk, R, heads, H // heads,
            dt_of(qkv), stream()))
        return dqkv
    check(lib_for(qkv, out, dout, dqkv).stdiff_attention_bwd(ptr(qkv), qkv.stride(0), ptr(out), out.stride(0), ptr(dout), dout.stride(0),
                                     ptr(lse), ptr(delta), ptr(dqkv), dqkv.stride(0), ptr(cu_rows),
                                     cu_rows.numel() - 1, max_chunk, R, heads, H // heads, dt_of(qkv), stream()))
    return dqkv


def ln_modulate_bwd(dy, x, mean, rstd, scale, dx, dshift, dscale, accumulate=True, gate=None, branch=None, dbranch=None,
                    dgate=None, split=None):
    """LayerNorm + modulate backward; with gate / branch / dbranch / dgate also the gated-residual backward of the
    branch below it on the finished dx row (dbranch = dx * gate, dgate = dx * branch): one launch instead of two."""
    require_gpu(dy, x, mean, rstd, scale, dx, dshift, dscale, gate, branch, dbranch, dgate)
    R, H = x.shape
    assert dshift.stride(0) == dscale.stride(0)
    lb = lib_for(dy, scale, dshift)
    if split is not None and split.slices > 1:       # dy = the partial products summed (and rounded to the dtype)
        z = lambda t_: t_.stride(0) if t_ is not None else 0
        check(lb.stdiff_ln_modulate_bwd_split(C.byref(split), dt_of(scale), ptr(x), x.stride(0), ptr(mean), ptr(rstd),
                                              ptr(scale), scale.stride(0), ptr(dx), dx.stride(0), int(accumulate),
                                              ptr(dshift), ptr(dscale), dshift.stride(0), ptr(gate), z(gate), ptr(branch),
                                              z(branch), ptr(dbranch), z(dbranch), ptr(dgate), z(dgate), R, H, stream()))
        return
    if gate is None:
        check(lb.stdiff_ln_modulate_bwd(ptr(dy), dy.stride(0), dt_of(dy), ptr(x), x.stride(0), ptr(mean), ptr(rstd),
                                        ptr(scale), scale.stride(0), ptr(dx), dx.stride(0), int(accumulate),
                                        ptr(dshift), ptr(dscale), dshift.stride(0), R, H, stream()))
        return
    check(lb.stdiff_ln_modulate_bwd_gate(ptr(dy), dy.stride(0), dt_of(dy), ptr(x), x.stride(0), ptr(mean), ptr(rstd),
                                         ptr(scale), scale.stride(0), ptr(dx), dx.stride(0), int(accumulate),
                                         ptr(dshift), ptr(dscale), dshift.stride(0), ptr(gate), gate.stride(0),
                                         ptr(branch), branch.stride(0), ptr(dbranch), dbranch.stride(0), ptr(dgate),
                                         dgate.stride(0), R, H, stream()))


def gate_bwd(dh, gate, branch, dbranch, dgate):
    require_gpu(dh, gate, branch, dbranch, dgate)
    R, H = dh.shape
    check(lib_for(gate, branch, dbranch, dgate).stdiff_gate_bwd(ptr(dh), dh.stride(0), ptr(gate), gate.stride(0), ptr(branch), branch.stride(0),
                                ptr(dbranch), dbranch.stride(0), ptr(dgate), dgate.stride(0), dt_of(gate), R, H,
                                stream()))


def gelu_bwd(dg, u_pre, du):
    require_gpu(dg, u_pre, du)
    R, Cc = dg.shape
    assert dg.stride(0) == u_pre.stride(0) == du.stride(0)
    check(lib_for(dg, u_pre, du).stdiff_gelu_bwd(ptr(dg), ptr(u_pre), ptr(du), dg.stride(0), dt_of(dg), R, Cc, stream()))


def silu_bwd(ds, c, dc, accumulate=False):
    require_gpu(ds, c, dc)
    R, H = ds.shape
    check(lib_for(ds, c, dc).stdiff_silu_bwd(ptr(ds), ds.stride(0), dt_of(ds), ptr(c), c.stride(0), dt_of(c), ptr(dc),
                                dc.stride(0), dt_of(dc), int(accumulate), R, H, stream()))


def colsum(dY, db, accumulate=False):
    require_gpu(dY, db)
    R, Cc = dY.shape
    check(lib_for(dY).stdiff_colsum(ptr(dY), dY.stride(0), dt_of(dY), ptr(db), int(accumulate), R, Cc, stream()))


def grad_sumsq(g, sumsq):
    require_gpu(g, sumsq)
    check(lib().stdiff_grad_sumsq(ptr(g), g.numel(), ptr(sumsq), stream()))


def adamw_args(p, g, m, v, sumsq, step_dev, scal, lr, max_norm=1.0, grad_scale=1.0, betas=(0.9, 0.999), eps=1e-8,
               weight_decay=0.0, bf16_mirror=None):
    """stdiff_adamw_args over the flat buffers (for adamw_prepare / adamw_range: the update issued range by range)."""
    require_gpu(p, g, m, v, sumsq, step_dev, scal, bf16_mirror)
    a = L.AdamWArgs()
    a.params, a.grads, a.exp_avg, a.exp_avg_sq, a.n = ptr(p), ptr(g), ptr(m), ptr(v), p.numel()
    a.sumsq, a.step_dev, a.scal = ptr(sumsq), ptr(step_dev), ptr(scal)
    a.max_norm, a.grad_scale, a.lr = max_norm, grad_scale, lr
    a.beta1, a.beta2, a.eps, a.weight_decay = betas[0], betas[1], eps, weight_decay
    if bf16_mirror is not None:
        assert bf16_mirror.dtype == torch.bfloat16 and bf16_mirror.numel() == p.numel()
        a.bf16_mirror = ptr(bf16_mirror)
    return a


def adamw_prepare(a):
    check(lib().stdiff_adamw_prepare(C.byref(a), stream()))


def adamw_range(a, lo, count):
    check(lib().stdiff_adamw_range(C.byref(a), lo, count, stream()))


def clip_adamw(p, g, m, v, sumsq, step_dev, scal, lr, max_norm=1.0, grad_scale=1.0, betas=(0.9, 0.999), eps=1e-8,
               weight_decay=0.0, bf16_mirror=None):
    require_gpu(p, g, m, v, sumsq, step_dev, scal, bf16_mirror)
    a = L.AdamWArgs()
    a.params, a.grads, a.exp_avg, a.exp_avg_sq, a.n = ptr(p), ptr(g), ptr(m), ptr(v), p.numel()
    a.sumsq, a.step_dev, a.scal = ptr(sumsq), ptr(step_dev), ptr(scal)
    a.max_norm, a.grad_scale, a.lr = max_norm, grad_scale, lr
    a.beta1, a.beta2, a.eps, a.weight_decay = betas[0], betas[1], eps, weight_decay
    if bf16_mirror is not None:
        assert bf16_mirror.dtype == torch.bfloat16 and bf16_mirror.numel() == p.numel()
        a.bf16_mirror = ptr(bf16_mirror)
    check(lib().stdiff_clip_adamw(C.byref(a), stream()))


class Graph:
    """hipGraph capture of everything enqueued on the current stream between begin() and end()."""

    def __init__(self):
        self.exec = None

    def begin(self):
        check(lib().stdiff_graph_begin(stream()))

    def end(self):
        h = C.c_void_p()
        check(lib().stdiff_graph_end(stream(), C.byref(h)))
        self.exec = h

    def launch(self):
        check(lib().stdiff_graph_launch(self.exec, stream()))

    def __del__(self):
        if self.exec is not None:
            try:
                lib().stdiff_graph_destroy(self.exec)
            except Exception:
                pass
            self.exec = None
