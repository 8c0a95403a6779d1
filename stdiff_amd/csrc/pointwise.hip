// Bandwidth-bound fused kernels of the denoiser: adaLN LayerNorm+modulate, SiLU of the conditioning vector,
// sinusoidal timestep embedding, operand packing.  One 64-lane wave owns one row; rows live in registers between
// the two reduction passes, every global access is a 16-byte (fp32) / 8-byte (bf16) vector per lane.
#include "common.h"

namespace {

// ---------------------------------------------------------------- LayerNorm + modulate (stDiff_model.py:41-44,96,99,126)
// NV = float4 vectors per lane: H <= 256*NV.
// SPLIT: x is not read but FORMED from a GEMM result that still lies as split-K partial products (stdiff_split_result):
// v = sum of the slices + bias, x = res + gate * v (no gate: x = v); x and v are written for the backward pass.  The
// arithmetic and its order are those of splitk_reduce_kernel's gated-residual epilogue, so nothing changes but the launch.
struct LnSplit {
    const float* ws; int ldws, slices; long long stride;
    const float* bias; const void* gate; int ldgate; const float* res; int ldres;
    float* x_out; int ldxo; void* v_out; int ldv;
};
template <int NV, typename TM, typename TO, bool SPLIT = false>
__global__ __launch_bounds__(256) void ln_modulate_kernel(const float* __restrict__ x, int ldx, const TM* shift,
                                                          const TM* scale, int ldmod, TO* y, int ldy, float* mean_out,
                                                          float* rstd_out, int R, int H, LnSplit sp) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    const float* xr = x + (size_t)row * ldx;
    f32x4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        int c = (i * 64 + lane) * 4;
        if constexpr (SPLIT) {
            v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (c < H) {
                const float* w = sp.ws + (size_t)row * sp.ldws + c;
                f32x4 a = sum_slices(w, sp.slices, sp.stride);
                if (sp.bias) a += *(const f32x4*)(sp.bias + c);
                if (sp.v_out) st4((TM*)sp.v_out + (size_t)row * sp.ldv + c, a);
                if (sp.gate) {
                    const f32x4 g = ld4((const TM*)sp.gate + (size_t)row * sp.ldgate + c);
                    const f32x4 r = *(const f32x4*)(sp.res + (size_t)row * sp.ldres + c);
#pragma unroll
                    for (int j = 0; j < 4; ++j) a[j] = r[j] + g[j] * a[j];
                }
                *(f32x4*)(sp.x_out + (size_t)row * sp.ldxo + c) = a;
                v[i] = a;
            }
        } else {
            v[i] = c < H ? *(const f32x4*)(xr + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        s += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    const float mean = wave_sum(s) / (float)H;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        int c = (i * 64 + lane) * 4;
        if (c < H) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { float d = v[i][j] - mean; q += d * d; }
        }
    }
    const float m2 = wave_sum(q);
    const float rstd = rsqrtf(m2 / (float)H + 1e-6f);
    if (y == nullptr) {                             // statistics only (stdiff_row_stats): (mean, M2) pairs
        if (lane == 0) { mean_out[2 * row] = mean; mean_out[2 * row + 1] = m2; }
        return;
    }
    if (mean_out && lane == 0) { mean_out[row] = mean; rstd_out[row] = rstd; }
    // (requesting shift / scale together with x, ahead of the two reductions, was slower: 35 against 31.5 us at
    // 20 000 x 1024 -- the 16 extra live registers cost more occupancy than the hidden latency is worth)
    const TM* sh = shift + (size_t)row * ldmod;
    const TM* sc = scale + (size_t)row * ldmod;
    TO* yr = y + (size_t)row * ldy;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        int c = (i * 64 + lane) * 4;
        if (c < H) {
            f32x4 a = ld4(sh + c), b = ld4(sc + c), o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = ln_mod_apply(v[i][j], mean, rstd, b[j], a[j]);
            st4(yr + c, o);
        }
    }
}

template <typename TM, typename TO>
int launch_ln(const float* x, int ldx, const void* shift, const void* scale, int ldmod, void* y, int ldy, float* mean,
              float* rstd, int R, int H, hipStream_t st, const LnSplit* sp = nullptr) {
    dim3 grid(cdiv(R, 4)), block(256);
#define LN_CASE(NV)                                                                                              \
    do {                                                                                                         \
        if (sp)                                                                                                  \
            hipLaunchKernelGGL((ln_modulate_kernel<NV, TM, TO, true>), grid, block, 0, st, x, ldx, (const TM*)shift, \
                               (const TM*)scale, ldmod, (TO*)y, ldy, mean, rstd, R, H, *sp);                     \
        else                                                                                                     \
            hipLaunchKernelGGL((ln_modulate_kernel<NV, TM, TO, false>), grid, block, 0, st, x, ldx, (const TM*)shift, \
                               (const TM*)scale, ldmod, (TO*)y, ldy, mean, rstd, R, H, LnSplit{});               \
    } while (0)
    if (H <= 256) LN_CASE(1);
    else if (H <= 512) LN_CASE(2);
    else if (H <= 1024) LN_CASE(4);
    else if (H <= 2048) LN_CASE(8);
    else if (H <= 4096) LN_CASE(16);
    else return STDIFF_ERR_SHAPE;
#undef LN_CASE
    STDIFF_LAUNCH_STATUS();
    return STDIFF_OK;
}

// ---------------------------------------------------------------- s = silu(temb[t] + ycond)
template <typename TO>
__global__ __launch_bounds__(256) void silu_cond_kernel(const float* __restrict__ temb, const int32_t* t_index,
                                                        const float* __restrict__ ycond, int ldy, TO* s, int lds,
                                                        int R, int H) {
    // block = 4 rows at a time (one wave per row), lanes over the row's float4 pieces: no per-element 64-bit
    // division (the flat grid-stride form spent more issue slots on i / (H/4), i % (H/4) than on the SiLU)
    const float* te = temb ? temb + (size_t)(*t_index) * H : nullptr;
    const int lane = threadIdx.x & 63;
    for (int r = blockIdx.x * 4 + (threadIdx.x >> 6); r < R; r += gridDim.x * 4) {
        const float* yr = ycond + (size_t)r * ldy;
        TO* sr = s + (size_t)r * lds;
        for (int c = lane * 4; c < H; c += 256) {
            f32x4 v = *(const f32x4*)(yr + c);
            if (te) {
                f32x4 t = *(const f32x4*)(te + c);
                v += t;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = silu_f(v[j]);
            st4(sr + c, v);
        }
    }
}

// ---------------------------------------------------------------- sinusoidal embedding (stDiff_model.py:61-80)
template <typename TO>
__global__ __launch_bounds__(256) void timestep_embedding_kernel(const int64_t* t, const float* freqs, TO* out,
                                                                 int ldo, int R) {
    // 256 threads = the 256 output columns of one row: [cos(t f_k) | sin(t f_k)]; f_k = exp(-ln(1e4) k / 128) comes
    // from the host (computed with the reference's own fp32 op sequence: t*f_k reaches ~1e3 rad, so an ulp of f_k matters)
    const int col = threadIdx.x, k = col & 127;
    const float f = freqs[k];
    for (int r = blockIdx.x; r < R; r += gridDim.x) {
        float a = (float)t[r] * f;
        Elem<TO>::st(out + (size_t)r * ldo + col, col < 128 ? cosf(a) : sinf(a));
    }
}

template <typename TO>
__global__ __launch_bounds__(256) void pack_rows_kernel(const float* __restrict__ src, int lds, TO* dst, int ldd, int R,
                                                        int C, int Cpad) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)R * Cpad;
         i += (size_t)gridDim.x * blockDim.x) {
        int r = (int)(i / Cpad), c = (int)(i % Cpad);
        Elem<TO>::st(dst + (size_t)r * ldd + c, c < C ? src[(size_t)r * lds + c] : 0.f);
    }
}

// dst[r, :] = src[idx[r], :] for two row-aligned matrices at once (the (x, x_cond) pair of a shuffled mini-batch):
// one wave per output row, 16-byte pieces when both rows allow it.  idx outside [0, n_src) is the caller's bug: the
// host wrapper checks the permutation it builds.
__global__ __launch_bounds__(256) void gather_rows2_kernel(const float* __restrict__ a, int lda, const float* __restrict__ b,
                                                           int ldb, const int64_t* __restrict__ idx, float* da, int ldda,
                                                           float* db, int lddb, int R, int C, int vec) {
    const int lane = threadIdx.x & 63;
    for (int r = blockIdx.x * 4 + (threadIdx.x >> 6); r < R; r += gridDim.x * 4) {
        const size_t s = (size_t)idx[r];
        const float* sa = a + s * lda;
        const float* sb = b ? b + s * ldb : nullptr;
        float* oa = da + (size_t)r * ldda;
        float* ob = db + (size_t)r * lddb;
        if (vec) {
            for (int c = lane * 4; c < C; c += 256) {
                *(f32x4*)(oa + c) = *(const f32x4*)(sa + c);
                if (sb) *(f32x4*)(ob + c) = *(const f32x4*)(sb + c);
            }
        } else {
            for (int c = lane; c < C; c += 64) {
                oa[c] = sa[c];
                if (sb) ob[c] = sb[c];
            }
        }
    }
}

// Expression pre-processing of the reference driver, fused per cell (test-stDiff.py:47-55; process/data.py:79-100):
//   [data_augment(fixed): + add_const]  ->  scanpy normalize_total(target_sum)  ->  log1p  ->  MaxAbsScaler over the
//   cell's genes (scale() transposes, so the "feature" is the cell).  Zero-count cells divide by 1, as in scanpy /
//   sklearn.  One wave per row, three sweeps (sum, max, write); the row is re-read from L2.
// SCALE_ONLY: just scale() (process/data.py:79-83) on an already normalised matrix: y = x / max|x| per cell, a true
// fp32 division like MaxAbsScaler's `X /= scale_` (bit-equal to sklearn on fp32 input; golden G10).
// log1p(z) for z >= 0 through the hardware log2: log(u) * z / (u - 1) with u = fl(1 + z) compensates the rounding of
// 1 + z (Kahan's form), so small z keep their relative accuracy; |error| <= 2 ulp of the result (v_log_f32: 1 ulp).
__device__ __forceinline__ float log1p_hw(float z) {
    const float u = 1.0f + z, d = u - 1.0f;
    const float l = __builtin_amdgcn_logf(u) * 0.69314718055994530942f;
    return d == 0.f ? z : l * (z / d);
}

template <bool SCALE_ONLY>
__global__ __launch_bounds__(256) void preprocess_rows_kernel(const float* __restrict__ x, int ldx, float* y, int ldy,
                                                              int R, int G, float add_const, float target_sum) {
    const int lane = threadIdx.x & 63;
    for (int r = blockIdx.x * 4 + (threadIdx.x >> 6); r < R; r += gridDim.x * 4) {
        const float* xr = x + (size_t)r * ldx;
        float k = 1.f;
        if constexpr (!SCALE_ONLY) {
            float s = 0.f;
            for (int c = lane; c < G; c += 64) s += xr[c] + add_const;
            s = wave_sum(s);
            k = target_sum / (s == 0.f ? 1.f : s);
        }
        auto value = [&](int c) { return SCALE_ONLY ? xr[c] : log1p_hw((xr[c] + add_const) * k); };
        float mx = 0.f;
        for (int c = lane; c < G; c += 64) mx = fmaxf(mx, fabsf(value(c)));
        mx = wave_max(mx);
        const float den = mx == 0.f ? 1.f : mx;
        float* yr = y + (size_t)r * ldy;
        for (int c = lane; c < G; c += 64) yr[c] = value(c) / den;
    }
}

// The same with the row kept in registers (G <= 256 * NV, 16-byte aligned rows): one read, one log1p per element, one
// write -- the three-sweep form above evaluates log1p twice per element and re-reads the row twice through 4-byte loads
// (1.1 TB/s at 20 000 x 1000; this form is bound by the two HBM passes).  Same per-element arithmetic; the row sum is
// taken in another order (last bits of the normalisation factor), scale() alone is bit-identical.
template <bool SCALE_ONLY, int NV>
__global__ __launch_bounds__(256) void preprocess_rows_reg_kernel(const float* __restrict__ x, int ldx, float* y, int ldy,
                                                                  int R, int G, float add_const, float target_sum) {
    const int lane = threadIdx.x & 63;
    for (int r = blockIdx.x * 4 + (threadIdx.x >> 6); r < R; r += gridDim.x * 4) {
        const float* xr = x + (size_t)r * ldx;
        f32x4 v[NV];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            v[i] = c < G ? *(const f32x4*)(xr + c) : f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (!SCALE_ONLY) {
                if (c < G) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { v[i][j] += add_const; s += v[i][j]; }
                }
            }
        }
        float mx = 0.f;
        if constexpr (!SCALE_ONLY) {
            s = wave_sum(s);
            const float k = target_sum / (s == 0.f ? 1.f : s);
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                const int c = (i * 64 + lane) * 4;
                if (c < G) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { v[i][j] = log1p_hw(v[i][j] * k); mx = fmaxf(mx, fabsf(v[i][j])); }
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < NV; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) mx = fmaxf(mx, fabsf(v[i][j]));
        }
        mx = wave_max(mx);
        const float den = mx == 0.f ? 1.f : mx;
        float* yr = y + (size_t)r * ldy;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            if (c < G) {
                f32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = v[i][j] / den;
                *(f32x4*)(yr + c) = o;
            }
        }
    }
}

inline int ew_grid(size_t n) {
    // grid-stride cap, swept 512 ... 16384 at 20 000 x 1024 (silu_cond: 34 / 28 / 25 / 24 / 25 us)
    static const int cap = [] { const char* e = getenv("STDIFF_EW_BLOCKS"); return e && atoi(e) > 0 ? atoi(e) : 256 * 16; }();
    return (int)std::min<size_t>((n + 255) / 256, (size_t)cap);
}
inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

extern "C" int stdiff_ln_modulate(const float* x, int ldx, const void* shift, const void* scale, int ldmod,
                                  int mod_dtype, void* y, int ldy, int out_dtype, float* mean, float* rstd, int R,
                                  int H, void* stream) {
    STDIFF_CHECK_ARG(x && shift && scale && y && R >= 0 && H > 0);
    STDIFF_CHECK_ARG((mean == nullptr) == (rstd == nullptr));
    if (R == 0) return STDIFF_OK;
    if (H % 4) return STDIFF_ERR_SHAPE;
    if (ldx % 4 || ldmod % 4 || ldy % 4 || !al16(x) || ((uintptr_t)shift & 7) || ((uintptr_t)scale & 7) ||
        ((uintptr_t)y & 7))
        return STDIFF_ERR_ALIGN;
    hipStream_t st = (hipStream_t)stream;
    const bool mb = mod_dtype == STDIFF_BF16, ob = out_dtype == STDIFF_BF16;
    if (mb && ob) return launch_ln<bf16_t, bf16_t>(x, ldx, shift, scale, ldmod, y, ldy, mean, rstd, R, H, st);
    if (!mb && !ob) return launch_ln<float, float>(x, ldx, shift, scale, ldmod, y, ldy, mean, rstd, R, H, st);
    if (!mb && ob) return launch_ln<float, bf16_t>(x, ldx, shift, scale, ldmod, y, ldy, mean, rstd, R, H, st);
    return launch_ln<bf16_t, float>(x, ldx, shift, scale, ldmod, y, ldy, mean, rstd, R, H, st);
}

extern "C" int stdiff_ln_modulate_split(const stdiff_split_result* sp, const float* bias, const void* gate, int ldgate,
                                        const float* res, int ldres, float* x, int ldx, void* v_out, int ldv,
                                        const void* shift, const void* scale, int ldmod, int mod_dtype, void* y, int ldy,
                                        int out_dtype, float* mean, float* rstd, int R, int H, void* stream) {
    STDIFF_CHECK_ARG(sp && sp->ws && sp->slices >= 1 && sp->slices <= STDIFF_MAX_SLICES && sp->ldws >= H && x && shift && scale && y && R >= 0 && H > 0);
    STDIFF_CHECK_ARG((mean == nullptr) == (rstd == nullptr) && (gate == nullptr) == (res == nullptr));
    STDIFF_CHECK_ARG(mod_dtype == STDIFF_F32 || mod_dtype == STDIFF_BF16);
    if (R == 0) return STDIFF_OK;
    if (H % 4) return STDIFF_ERR_SHAPE;
    const bool mb = mod_dtype == STDIFF_BF16, ob = out_dtype == STDIFF_BF16;
    auto ok_mod = [&](const void* q) { return (((uintptr_t)q) & (mb ? 7 : 15)) == 0; };
    if (ldx % 4 || ldmod % 4 || ldy % 4 || sp->ldws % 4 || sp->slice_stride % 4 || !al16(sp->ws) || !al16(x) ||
        !ok_mod(shift) || !ok_mod(scale) || ((uintptr_t)y & 7) || (bias && !al16(bias)) ||
        (gate && (ldgate % 4 || ldres % 4 || !ok_mod(gate) || !al16(res))) || (v_out && (ldv % 4 || !ok_mod(v_out))))
        return STDIFF_ERR_ALIGN;
    LnSplit q;
    q.ws = sp->ws; q.ldws = sp->ldws; q.slices = sp->slices; q.stride = sp->slice_stride;
    q.bias = bias; q.gate = gate; q.ldgate = ldgate; q.res = res; q.ldres = ldres;
    q.x_out = x; q.ldxo = ldx; q.v_out = v_out; q.ldv = ldv;
    hipStream_t st = (hipStream_t)stream;
    if (mb && ob) return launch_ln<bf16_t, bf16_t>(x, ldx, shift, scale, ldmod, y, ldy, mean, rstd, R, H, st, &q);
    if (!mb && !ob) return launch_ln<float, float>(x, ldx, shift, scale, ldmod, y, ldy, mean, rstd, R, H, st, &q);
    if (!mb && ob) return launch_ln<float, bf16_t>(x, ldx, shift, scale, ldmod, y, ldy, mean, rstd, R, H, st, &q);
    return launch_ln<bf16_t, float>(x, ldx, shift, scale, ldmod, y, ldy, mean, rstd, R, H, st, &q);
}

extern "C" int stdiff_row_stats(const float* x, int ldx, float* parts, int R, int H, void* stream) {
    STDIFF_CHECK_ARG(x && parts && R >= 0 && H > 0);
    if (R == 0) return STDIFF_OK;
    if (H % 4) return STDIFF_ERR_SHAPE;
    if (ldx % 4 || !al16(x)) return STDIFF_ERR_ALIGN;
    // the same kernel as stdiff_ln_modulate (same reduction order, same bits) with the output pass switched off
    return launch_ln<bf16_t, bf16_t>(x, ldx, nullptr, nullptr, 0, nullptr, 0, parts, nullptr, R, H, (hipStream_t)stream);
}

extern "C" int stdiff_silu_cond(const float* temb, const int32_t* t_index, const float* ycond, int ldy, void* s,
                                int lds, int out_dtype, int R, int H, void* stream) {
    STDIFF_CHECK_ARG(ycond && s && R >= 0 && H > 0 && (temb == nullptr || t_index != nullptr));
    if (R == 0) return STDIFF_OK;
    if (H % 4) return STDIFF_ERR_SHAPE;
    if (ldy % 4 || lds % 4 || !al16(ycond) || (temb && !al16(temb)) || ((uintptr_t)s & 7)) return STDIFF_ERR_ALIGN;
    dim3 grid((unsigned)std::min<size_t>(((size_t)R + 3) / 4, (size_t)16384)), block(256);
    if (out_dtype == STDIFF_BF16)
        hipLaunchKernelGGL(silu_cond_kernel<bf16_t>, grid, block, 0, (hipStream_t)stream, temb, t_index, ycond, ldy,
                           (bf16_t*)s, lds, R, H);
    else
        hipLaunchKernelGGL(silu_cond_kernel<float>, grid, block, 0, (hipStream_t)stream, temb, t_index, ycond, ldy,
                           (float*)s, lds, R, H);
    STDIFF_LAUNCH_STATUS();
    return STDIFF_OK;
}

extern "C" int stdiff_timestep_embedding(const int64_t* t, const float* freqs, void* out, int ldo, int out_dtype, int R,
                                         void* stream) {
    STDIFF_CHECK_ARG(t && freqs && out && R >= 0 && ldo >= 256);
    if (R == 0) return STDIFF_OK;
    dim3 grid(std::min(R, 4096)), block(256);
    if (out_dtype == STDIFF_BF16)
        hipLaunchKernelGGL(timestep_embedding_kernel<bf16_t>, grid, block, 0, (hipStream_t)stream, t, freqs, (bf16_t*)out, ldo, R);
    else
        hipLaunchKernelGGL(timestep_embedding_kernel<float>, grid, block, 0, (hipStream_t)stream, t, freqs, (float*)out, ldo, R);
    STDIFF_LAUNCH_STATUS();
    return STDIFF_OK;
}

extern "C" int stdiff_pack_rows(const float* src, int lds, void* dst, int ldd, int dtype, int R, int C, int Cpad,
                                void* stream) {
    STDIFF_CHECK_ARG(src && dst && R >= 0 && C > 0 && Cpad >= C && ldd >= Cpad && lds >= C);
    if (R == 0) return STDIFF_OK;
    dim3 grid(ew_grid((size_t)R * Cpad)), block(256);
    if (dtype == STDIFF_BF16)
        hipLaunchKernelGGL(pack_rows_kernel<bf16_t>, grid, block, 0, (hipStream_t)stream, src, lds, (bf16_t*)dst, ldd, R, C, Cpad);
    else
        hipLaunchKernelGGL(pack_rows_kernel<float>, grid, block, 0, (hipStream_t)stream, src, lds, (float*)dst, ldd, R, C, Cpad);
    STDIFF_LAUNCH_STATUS();
    return STDIFF_OK;
}

extern "C" int stdiff_gather_rows2(const float* a, int lda, const float* b, int ldb, const int64_t* idx, float* da,
                                   int ldda, float* db, int lddb, int R, int C, void* stream) {
    STDIFF_CHECK_ARG(R >= 0 && C > 0);
    if (R == 0) return STDIFF_OK;
    STDIFF_CHECK_ARG(a && idx && da && lda >= C && ldda >= C);
    STDIFF_CHECK_ARG(!b || (db && ldb >= C && lddb >= C));
    int vec = C % 4 == 0 && lda % 4 == 0 && ldda % 4 == 0 && al16(a) && al16(da);
    if (b) vec = vec && ldb % 4 == 0 && lddb % 4 == 0 && al16(b) && al16(db);
    hipLaunchKernelGGL(gather_rows2_kernel, dim3(std::min(cdiv(R, 4), 4096)), dim3(256), 0, (hipStream_t)stream, a, lda, b,
                       ldb, idx, da, ldda, db, lddb, R, C, vec);
    STDIFF_LAUNCH_STATUS();
    return STDIFF_OK;
}

extern "C" int stdiff_preprocess_rows(const float* x, int ldx, float* y, int ldy, int R, int G, float add_const,
                                      float target_sum, void* stream) {
    STDIFF_CHECK_ARG(R >= 0 && G > 0 && target_sum >= 0.f);
    if (R == 0) return STDIFF_OK;
    STDIFF_CHECK_ARG(x && y && ldx >= G && ldy >= G);
    const dim3 grid(std::min(cdiv(R, 4), 8192)), block(256);
    const bool scale_only = target_sum == 0.f;
    if (G % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0 && G <= 4096) {
#define PRE_CASE(NV)                                                                                               \
    do {                                                                                                           \
        if (scale_only)                                                                                            \
            hipLaunchKernelGGL((preprocess_rows_reg_kernel<true, NV>), grid, block, 0, (hipStream_t)stream, x, ldx, y, ldy, \
                               R, G, 0.f, 0.f);                                                                    \
        else                                                                                                       \
            hipLaunchKernelGGL((preprocess_rows_reg_kernel<false, NV>), grid, block, 0, (hipStream_t)stream, x, ldx, y, \
                               ldy, R, G, add_const, target_sum);                                                  \
    } while (0)
        if (G <= 256) PRE_CASE(1);
        else if (G <= 512) PRE_CASE(2);
        else if (G <= 1024) PRE_CASE(4);
        else if (G <= 2048) PRE_CASE(8);
        else PRE_CASE(16);
#undef PRE_CASE
        STDIFF_LAUNCH_STATUS();
        return STDIFF_OK;
    }
    if (target_sum == 0.f)         // scale() alone: no normalize_total / log1p (add_const is ignored)
        hipLaunchKernelGGL(preprocess_rows_kernel<true>, grid, block, 0, (hipStream_t)stream, x, ldx, y, ldy, R, G, 0.f, 0.f);
    else
        hipLaunchKernelGGL(preprocess_rows_kernel<false>, grid, block, 0, (hipStream_t)stream, x, ldx, y, ldy, R, G,
                           add_const, target_sum);
    STDIFF_LAUNCH_STATUS();
    return STDIFF_OK;
}
