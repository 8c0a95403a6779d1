// Backward-pass and optimizer kernels of normal_train_stDiff (stDiff_train.py:56-81); all HBM-bound.
//   ln_modulate_bwd : d(LayerNorm*(1+scale)+shift)   -> dx (fp32), dshift, dscale
//   gate_bwd        : d(x + gate*branch)             -> dbranch, dgate
//   gelu_bwd        : erf-GELU derivative
//   silu_bwd        : SiLU derivative (conditioning vector and time MLP)
//   colsum          : bias gradients
//   grad_sumsq / clip_adamw : clip_grad_norm_(1.0) + AdamW(weight_decay) over ONE flat fp32 parameter buffer
#include "common.h"

namespace {

template <typename T> __device__ __forceinline__ float ldf(const void* p, size_t i) { return Elem<T>::ld((const T*)p + i); }

// ---------------------------------------------------------------- LayerNorm-modulate backward, one wave per row
// y = xhat*(1+scale)+shift, xhat = (x-mean)*rstd:  g = dy*(1+scale);  dx = rstd*(g - mean(g) - xhat*mean(g*xhat))
// SPLIT: dy is not read but summed from split-K partial products (stdiff_split_result) and rounded to T, which is what
// the reduce launch it replaces would have stored.
template <int NV, typename T, bool SPLIT = false>
__global__ __launch_bounds__(256) void ln_mod_bwd_kernel(const T* dy, int lddy, const float* x, int ldx,
                                                         const float* mean, const float* rstd, const T* scale,
                                                         int ldmod, float* dx, int lddx, int accumulate, T* dshift,
                                                         T* dscale, int lddmod, int R, int H, const T* gate,
                                                         int ldgate, const T* branch, int ldbr, T* dbranch, int lddb,
                                                         T* dgate, int lddg, const float* ws, int ldws, int slices,
                                                         long long stride) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    const float mu = mean[row], rs = rstd[row];
    f32x4 g[NV], xh[NV];
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        int c = (i * 64 + lane) * 4;
        if (c < H) {
            f32x4 d;
            if constexpr (SPLIT) {
                const float* w = ws + (size_t)row * ldws + c;
                d = sum_slices(w, slices, stride);
                if constexpr (sizeof(T) == 2) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) d[j] = bf2f(f2bf(d[j]));
                }
            } else {
                d = ld4(dy + (size_t)row * lddy + c);
            }
            f32x4 xv = *(const f32x4*)(x + (size_t)row * ldx + c), sc = ld4(scale + (size_t)row * ldmod + c), ds;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                xh[i][j] = (xv[j] - mu) * rs;
                g[i][j] = d[j] * (1.0f + sc[j]);
                ds[j] = d[j] * xh[i][j];
                sg += g[i][j];
                sgx += g[i][j] * xh[i][j];
            }
            st4(dshift + (size_t)row * lddmod + c, d);
            st4(dscale + (size_t)row * lddmod + c, ds);
        } else {
            g[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            xh[i] = g[i];
        }
    }
    const float mg = wave_sum(sg) / (float)H, mgx = wave_sum(sgx) / (float)H;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        int c = (i * 64 + lane) * 4;
        if (c < H) {
            float* dp = dx + (size_t)row * lddx + c;
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = rs * (g[i][j] - mg - xh[i][j] * mgx);
            if (accumulate) {
                f32x4 old = *(const f32x4*)dp;
                o += old;
            }
            *(f32x4*)dp = o;
            // the gated residual that FOLLOWS in the backward chain (x_next = x + gate * branch, d x_next = the dx just
            // formed): dbranch = dx * gate, dgate = dx * branch -- stdiff_gate_bwd's arithmetic on the value it would
            // have re-read, without its launch
            if (gate) {
                const f32x4 gv = ld4(gate + (size_t)row * ldgate + c), bv = ld4(branch + (size_t)row * ldbr + c);
                st4(dbranch + (size_t)row * lddb + c, o * gv);
                st4(dgate + (size_t)row * lddg + c, o * bv);
            }
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void gate_bwd_kernel(const float* dh, int lddh, const T* gate, int ldgate,
                                                       const T* branch, int ldbr, T* dbranch, int lddb, T* dgate,
                                                       int lddg, int R, int H) {
    const int hv = H >> 2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)R * hv;
         i += (size_t)gridDim.x * blockDim.x) {
        int r = (int)(i / hv), c = (int)(i % hv) * 4;
        f32x4 d = *(const f32x4*)(dh + (size_t)r * lddh + c), g = ld4(gate + (size_t)r * ldgate + c),
              b = ld4(branch + (size_t)r * ldbr + c);
        st4(dbranch + (size_t)r * lddb + c, d * g);
        st4(dgate + (size_t)r * lddg + c, d * b);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const T* dg, const T* u, T* du, int ld, int R, int C) {
    const int cv = C >> 2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)R * cv;
         i += (size_t)gridDim.x * blockDim.x) {
        int r = (int)(i / cv), c = (int)(i % cv) * 4;
        f32x4 d = ld4(dg + (size_t)r * ld + c), v = ld4(u + (size_t)r * ld + c), o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float x = v[j];
            float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
            float pdf = 0.3989422804014327f * expf(-0.5f * x * x);
            o[j] = d[j] * (cdf + x * pdf);
        }
        st4(du + (size_t)r * ld + c, o);
    }
}

template <typename TD, typename TC, typename TO>
__global__ __launch_bounds__(256) void silu_bwd_kernel(const TD* ds, int ldds, const TC* c, int ldc, TO* dc, int lddc,
                                                       int accumulate, int R, int H) {
    const int hv = H >> 2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < (size_t)R * hv;
         i += (size_t)gridDim.x * blockDim.x) {
        int r = (int)(i / hv), col = (int)(i % hv) * 4;
        f32x4 d = ld4(ds + (size_t)r * ldds + col), v = ld4(c + (size_t)r * ldc + col), o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float sg = 1.0f / (1.0f + expf(-v[j]));
            o[j] = d[j] * sg * (1.0f + v[j] * (1.0f - sg));
        }
        TO* out = dc + (size_t)r * lddc + col;
        if (accumulate) {
            f32x4 old = ld4(out);
            o += old;
        }
        st4(out, o);
    }
}

// db[c] (+)= sum_r dY[r,c]: block = 64 columns x 4 row groups
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* dY, int ld, float* db, int accumulate, int R, int C) {
    __shared__ float part[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    float s = 0.f;
    if (c < C)
        for (int r = g; r < R; r += 4) s += Elem<T>::ld(dY + (size_t)r * ld + c);
    part[g][threadIdx.x & 63] = s;
    __syncthreads();
    if (g == 0 && c < C) {
        float t = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
        db[c] = accumulate ? db[c] + t : t;
    }
}

// 4-wide variant: a block owns 32 columns (8 lanes x 4) and 32 row slots; every lane streams its rows with independent
// 8/16-byte loads (the scalar kernel above spent 25 us per bias on 128 dependent 2-byte loads), then one LDS tree.
// Fixed summation order: deterministic.
template <typename T>
__global__ __launch_bounds__(256) void colsum4_kernel(const T* dY, int ld, float* db, int accumulate, int R, int C) {
    __shared__ f32x4 part[32][8];
    const int cg = threadIdx.x & 7, slot = threadIdx.x >> 3;
    const int c = blockIdx.x * 32 + cg * 4;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
    if (c < C) {
        int r = slot;
        for (; r + 32 < R; r += 64) {
            s0 += ld4(dY + (size_t)r * ld + c);
            s1 += ld4(dY + (size_t)(r + 32) * ld + c);
        }
        if (r < R) s0 += ld4(dY + (size_t)r * ld + c);
    }
    part[slot][cg] = s0 + s1;
    __syncthreads();
    for (int h = 16; h > 0; h >>= 1) {
        if (slot < h) part[slot][cg] += part[slot + h][cg];
        __syncthreads();
    }
    if (slot == 0 && c < C) {
        f32x4 t = part[0][cg];
        if (accumulate) t += *(const f32x4*)(db + c);
        *(f32x4*)(db + c) = t;
    }
}

// ---------------------------------------------------------------- optimizer over a flat buffer
__global__ __launch_bounds__(256) void sumsq_kernel(const float* g, int64_t n, float* out) {
    float acc = 0.f;
    const int64_t nv = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
        f32x4 v = __builtin_nontemporal_load((const f32x4*)(g + 4 * i));
        acc += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int64_t i = nv * 4; i < n; ++i) acc += g[i] * g[i];
    acc = wave_sum(acc);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, (part[0] + part[1]) + (part[2] + part[3]));
}

// out (+)= sum of parts[0 .. n): one workgroup, thread t adds elements t, t + 1024, ... in order, then a fixed tree
__global__ __launch_bounds__(1024) void sum_parts_kernel(const float* parts, int64_t n, float* out, int accumulate) {
    float acc = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 1024) acc += parts[i];
    acc = wave_sum(acc);
    __shared__ float part[16];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += part[i];
        *out = accumulate ? *out + t : t;
    }
}

// torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW, one pass: 16 B read of p,g,m,v, 12 B write -> 28 B/param
__global__ void adam_prep_kernel(const float* sumsq, int32_t* step_dev, float* scal, float max_norm, float gscale,
                                 float lr, float b1, float b2) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int step = *step_dev + 1;
    *step_dev = step;
    const double bc1 = 1.0 - pow((double)b1, (double)step), bc2 = 1.0 - pow((double)b2, (double)step);
    const float norm = sqrtf(*sumsq) * gscale;
    scal[0] = fminf(max_norm / (norm + 1e-6f), 1.0f) * gscale;
    scal[1] = (float)((double)lr / bc1);
    scal[2] = (float)sqrt(bc2);
    scal[3] = norm;
}

// mirror (optional): bf16 copy of the updated parameters at the same flat offsets = the packed MFMA operands of the
// next forward, written here (+2 B/param) instead of by a separate repack pass per weight (read 4 + write 2).
template <bool NTM>
__global__ __launch_bounds__(256) void clip_adamw_kernel(float* p, const float* g, float* m, float* v, int64_t n,
                                                         const float* scal, float lr, float b1, float b2, float eps,
                                                         float wd, bf16_t* mirror) {
    const float coef = scal[0], step = scal[1], bc2_sqrt = scal[2];
    const int64_t nv = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (int64_t)gridDim.x * blockDim.x) {
        f32x4 pp, gg, mm, vv;
        if constexpr (NTM) {
            pp = __builtin_nontemporal_load((const f32x4*)(p + 4 * i));
            gg = __builtin_nontemporal_load((const f32x4*)(g + 4 * i));
            mm = __builtin_nontemporal_load((const f32x4*)(m + 4 * i));
            vv = __builtin_nontemporal_load((const f32x4*)(v + 4 * i));
        } else {
            pp = *(const f32x4*)(p + 4 * i); gg = *(const f32x4*)(g + 4 * i); mm = *(const f32x4*)(m + 4 * i);
            vv = *(const f32x4*)(v + 4 * i);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float gr = gg[j] * coef;
            pp[j] *= 1.0f - lr * wd;
            mm[j] = mm[j] * b1 + gr * (1.0f - b1);
            vv[j] = vv[j] * b2 + gr * gr * (1.0f - b2);
            pp[j] -= step * (mm[j] / (sqrtf(vv[j]) / bc2_sqrt + eps));
        }
        if constexpr (NTM) {
            __builtin_nontemporal_store(pp, (f32x4*)(p + 4 * i));
            __builtin_nontemporal_store(mm, (f32x4*)(m + 4 * i));
            __builtin_nontemporal_store(vv, (f32x4*)(v + 4 * i));
        } else {
            *(f32x4*)(p + 4 * i) = pp;
            *(f32x4*)(m + 4 * i) = mm;
            *(f32x4*)(v + 4 * i) = vv;
        }
        if (mirror) st4(mirror + 4 * i, pp);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int64_t i = nv * 4; i < n; ++i) {
            float gr = g[i] * coef;
            p[i] *= 1.0f - lr * wd;
            m[i] = m[i] * b1 + gr * (1.0f - b1);
            v[i] = v[i] * b2 + gr * gr * (1.0f - b2);
            p[i] -= step * (m[i] / (sqrtf(v[i]) / bc2_sqrt + eps));
            if (mirror) mirror[i] = f2bf(p[i]);
        }
}

inline int ew_grid(size_t n) { return (int)std::min<size_t>((n + 255) / 256, 256 * 8); }
inline bool al16(const void* p) { return p == nullptr || ((uintptr_t)p & 15) == 0; }
inline bool al8(const void* p) { return p == nullptr || ((uintptr_t)p & 7) == 0; }

template <typename T>
int launch_ln_bwd(const void* dy, int lddy, const float* x, int ldx, const float* mean, const float* rstd,
                  const void* scale, int ldmod, float* dx, int lddx, int acc, void* dshift, void* dscale, int lddmod,
                  int R, int H, hipStream_t st, const void* gate = nullptr, int ldgate = 0, const void* branch = nullptr,
                  int ldbr = 0, void* dbranch = nullptr, int lddb = 0, void* dgate = nullptr, int lddg = 0,
                  const stdiff_split_result* sp = nullptr) {
    dim3 grid(cdiv(R, 4)), block(256);
#define CASE(NV)                                                                                                   \
    do {                                                                                                           \
        if (sp)                                                                                                    \
            hipLaunchKernelGGL((ln_mod_bwd_kernel<NV, T, true>), grid, block, 0, st, (const T*)nullptr, 0, x, ldx, mean,   \
                               rstd, (const T*)scale, ldmod, dx, lddx, acc, (T*)dshift, (T*)dscale, lddmod, R, H,  \
                               (const T*)gate, ldgate, (const T*)branch, ldbr, (T*)dbranch, lddb, (T*)dgate, lddg, \
                               sp->ws, sp->ldws, sp->slices, sp->slice_stride);                                    \
        else                                                                                                       \
            hipLaunchKernelGGL((ln_mod_bwd_kernel<NV, T, false>), grid, block, 0, st, (const T*)dy, lddy, x, ldx, mean,    \
                               rstd, (const T*)scale, ldmod, dx, lddx, acc, (T*)dshift, (T*)dscale, lddmod, R, H,  \
                               (const T*)gate, ldgate, (const T*)branch, ldbr, (T*)dbranch, lddb, (T*)dgate, lddg, \
                               (const float*)nullptr, 0, 0, 0ll);                                                  \
    } while (0)
    if (H <= 256) CASE(1);
    else if (H <= 512) CASE(2);
    else if (H <= 1024) CASE(4);
    else if (H <= 2048) CASE(8);
    else if (H <= 4096) CASE(16);
    else return STDIFF_ERR_SHAPE;
#undef CASE
    STDIFF_LAUNCH_STATUS();
    return STDIFF_OK;
}

}  // namespace

extern "C" int stdiff_ln_modulate_bwd(const void* dy, int lddy, int dtype, const float* x, int ldx, const float* mean,
                                      const float* rstd, const void* scale, int ldmod, float* dx, int lddx,
                                      int accumulate, void* dshift, void* dscale, int lddmod, int R, int H,
                                      void* stream) {
    STDIFF_CHECK_ARG(dy && x && mean && rstd && scale && dx && dshift && dscale && R >= 0 && H > 0);
    STDIFF_CHECK_ARG(dtype == STDIFF_F32 || dtype == STDIFF_BF16);
    if (R == 0) return STDIFF_OK;
    if (H % 4) return STDIFF_ERR_SHAPE;
    if (lddy % 4 || ldx % 4 || ldmod % 4 || lddx % 4 || lddmod % 4 || !al16(x) || !al16(dx) || !al8(dy) || !al8(scale) ||
        !al8(dshift) || !al8(dscale))
        return STDIFF_ERR_ALIGN;
    if (dtype == STDIFF_BF16)
        return launch_ln_bwd<bf16_t>(dy, lddy, x, ldx, mean, rstd, scale, ldmod, dx, lddx, accumulate, dshift, dscale,
                                     lddmod, R, H, (hipStream_t)stream);
    return launch_ln_bwd<float>(dy, lddy, x, ldx, mean, rstd, scale, ldmod, dx, lddx, accumulate, dshift, dscale,
                                lddmod, R, H, (hipStream_t)stream);
}

extern "C" int stdiff_ln_modulate_bwd_gate(const void* dy, int lddy, int dtype, const float* x, int ldx,
                                           const float* mean, const float* rstd, const void* scale, int ldmod, float* dx,
                                           int lddx, int accumulate, void* dshift, void* dscale, int lddmod,
                                           const void* gate, int ldgate, const void* branch, int ldbr, void* dbranch,
                                           int lddb, void* dgate, int lddg, int R, int H, void* stream) {
    STDIFF_CHECK_ARG(dy && x && mean && rstd && scale && dx && dshift && dscale && R >= 0 && H > 0);
    STDIFF_CHECK_ARG(gate && branch && dbranch && dgate);
    STDIFF_CHECK_ARG(dtype == STDIFF_F32 || dtype == STDIFF_BF16);
    if (R == 0) return STDIFF_OK;
    if (H % 4) return STDIFF_ERR_SHAPE;
    if (lddy % 4 || ldx % 4 || ldmod % 4 || lddx % 4 || lddmod % 4 || !al16(x) || !al16(dx) || !al8(dy) || !al8(scale) ||
        !al8(dshift) || !al8(dscale) || ldgate % 4 || ldbr % 4 || lddb % 4 || lddg % 4 || !al8(gate) || !al8(branch) ||
        !al8(dbranch) || !al8(dgate))
        return STDIFF_ERR_ALIGN;
    if (dtype == STDIFF_F32 && (!al16(dy) || !al16(scale) || !al16(dshift) || !al16(dscale) || !al16(gate) ||
                                !al16(branch) || !al16(dbranch) || !al16(dgate)))
        return STDIFF_ERR_ALIGN;
    if (dtype == STDIFF_BF16)
        return launch_ln_bwd<bf16_t>(dy, lddy, x, ldx, mean, rstd, scale, ldmod, dx, lddx, accumulate, dshift, dscale,
                                     lddmod, R, H, (hipStream_t)stream, gate, ldgate, branch, ldbr, dbranch, lddb, dgate, lddg);
    return launch_ln_bwd<float>(dy, lddy, x, ldx, mean, rstd, scale, ldmod, dx, lddx, accumulate, dshift, dscale,
                                lddmod, R, H, (hipStream_t)stream, gate, ldgate, branch, ldbr, dbranch, lddb, dgate, lddg);
}

extern "C" int stdiff_ln_modulate_bwd_split(const stdiff_split_result* sp, int dtype, const float* x, int ldx,
                                            const float* mean, const float* rstd, const void* scale, int ldmod, float* dx,
                                            int lddx, int accumulate, void* dshift, void* dscale, int lddmod,
                                            const void* gate, int ldgate, const void* branch, int ldbr, void* dbranch,
                                            int lddb, void* dgate, int lddg, int R, int H, void* stream) {
    STDIFF_CHECK_ARG(sp && sp->ws && sp->slices >= 1 && sp->slices <= STDIFF_MAX_SLICES && sp->ldws >= H);
    STDIFF_CHECK_ARG(x && mean && rstd && scale && dx && dshift && dscale && R >= 0 && H > 0);
    STDIFF_CHECK_ARG(gate ? (branch && dbranch && dgate) : (!branch && !dbranch && !dgate));
    STDIFF_CHECK_ARG(dtype == STDIFF_F32 || dtype == STDIFF_BF16);
    if (R == 0) return STDIFF_OK;
    if (H % 4) return STDIFF_ERR_SHAPE;
    const bool bf = dtype == STDIFF_BF16;
    auto ok = [&](const void* q) { return bf ? al8(q) : al16(q); };
    if (sp->ldws % 4 || sp->slice_stride % 4 || !al16(sp->ws) || ldx % 4 || ldmod % 4 || lddx % 4 || lddmod % 4 ||
        !al16(x) || !al16(dx) || !ok(scale) || !ok(dshift) || !ok(dscale))
        return STDIFF_ERR_ALIGN;
    if (gate && (ldgate % 4 || ldbr % 4 || lddb % 4 || lddg % 4 || !ok(gate) || !ok(branch) || !ok(dbranch) || !ok(dgate)))
        return STDIFF_ERR_ALIGN;
    if (bf)
        return launch_ln_bwd<bf16_t>(nullptr, 0, x, ldx, mean, rstd, scale, ldmod, dx, lddx, accumulate, dshift, dscale,
                                     lddmod, R, H, (hipStream_t)stream, gate, ldgate, branch, ldbr, dbranch, lddb, dgate,
                                     lddg, sp);
    return launch_ln_bwd<float>(nullptr, 0, x, ldx, mean, rstd, scale, ldmod, dx, lddx, accumulate, dshift, dscale, lddmod,
                                R, H, (hipStream_t)stream, gate, ldgate, branch, ldbr, dbranch, lddb, dgate, lddg, sp);
}

extern "C" int stdiff_gate_bwd(const float* dh, int lddh, const void* gate, int ldgate, const void* branch, int ldbr,
                               void* dbranch, int lddb, void* dgate, int lddg, int dtype, int R, int H, void* stream) {
    STDIFF_CHECK_ARG(dh && gate && branch && dbranch && dgate && R >= 0 && H > 0);
    STDIFF_CHECK_ARG(dtype == STDIFF_F32 || dtype == STDIFF_BF16);
    if (R == 0) return STDIFF_OK;
    if (H % 4) return STDIFF_ERR_SHAPE;
    if (lddh % 4 || ldgate % 4 || ldbr % 4 || lddb % 4 || lddg % 4 || !al16(dh) || !al8(gate) || !al8(branch) ||
        !al8(dbranch) || !al8(dgate))
        return STDIFF_ERR_ALIGN;
    dim3 grid(ew_grid((size_t)R * H / 4)), block(256);
    if (dtype == STDIFF_BF16)
        hipLaunchKernelGGL(gate_bwd_kernel<bf16_t>, grid, block, 0, (hipStream_t)stream, dh, lddh, (const bf16_t*)gate,
                           ldgate, (const bf16_t*)branch, ldbr, (bf16_t*)dbranch, lddb, (bf16_t*)dgate, lddg, R, H);
    else
        hipLaunchKernelGGL(gate_bwd_kernel<float>, grid, block, 0, (hipStream_t)stream, dh, lddh, (const float*)gate,
                           ldgate, (const float*)branch, ldbr, (float*)dbranch, lddb, (float*)dgate, lddg, R, H);
    STDIFF_LAUNCH_STATUS();
    return STDIFF_OK;
}

extern "C" int stdiff_gelu_bwd(const void* dg, const void* u_pre, void* du, int ld, int dtype, int R, int C,
                               void* stream) {
    STDIFF_CHECK_ARG(dg && u_pre && du && R >= 0 && C > 0 && ld >= C);
    STDIFF_CHECK_ARG(dtype == STDIFF_F32 || dtype == STDIFF_BF16);
    if (R == 0) return STDIFF_OK;
    if (C % 4) return STDIFF_ERR_SHAPE;
    if (ld % 4 || !al8(dg) || !al8(u_pre) || !al8(du)) return STDIFF_ERR_ALIGN;
    dim3 grid(ew_grid((size_t)R * C / 4)), block(256);
    if (dtype == STDIFF_BF16)
        hipLaunchKernelGGL(gelu_bwd_kernel<bf16_t>, grid, block, 0, (hipStream_t)stream, (const bf16_t*)dg,
                           (const bf16_t*)u_pre, (bf16_t*)du, ld, R, C);
    else
        hipLaunchKernelGGL(gelu_bwd_kernel<float>, grid, block, 0, (hipStream_t)stream, (const float*)dg,
                           (const float*)u_pre, (float*)du, ld, R, C);
    STDIFF_LAUNCH_STATUS();
    return STDIFF_OK;
}

extern "C" int stdiff_silu_bwd(const void* ds, int ldds, int ds_dtype, const void* c, int ldc, int c_dtype, void* dc,
                               int lddc, int dc_dtype, int accumulate, int R, int H, void* stream) {
    STDIFF_CHECK_ARG(ds && c && dc && R >= 0 && H > 0);
    if (R == 0) return STDIFF_OK;
    if (H % 4) return STDIFF_ERR_SHAPE;
    if (ldds % 4 || ldc % 4 || lddc % 4 || !al8(ds) || !al8(c) || !al8(dc)) return STDIFF_ERR_ALIGN;
    if (ds_dtype == STDIFF_F32 && (!al16(ds))) return STDIFF_ERR_ALIGN;
    dim3 grid(ew_grid((size_t)R * H / 4)), block(256);
    hipStream_t st = (hipStream_t)stream;
    const int key = (ds_dtype == STDIFF_BF16) * 4 + (c_dtype == STDIFF_BF16) * 2 + (dc_dtype == STDIFF_BF16);
#define SB(TD, TC, TO)                                                                                       \
    hipLaunchKernelGGL((silu_bwd_kernel<TD, TC, TO>), grid, block, 0, st, (const TD*)ds, ldds, (const TC*)c, \
                       ldc, (TO*)dc, lddc, accumulate, R, H)
    switch (key) {
        case 0: SB(float, float, float); break;
        case 1: SB(float, float, bf16_t); break;
        case 2: SB(float, bf16_t, float); break;
        case 3: SB(float, bf16_t, bf16_t); break;
        case 4: SB(bf16_t, float, float); break;
        case 5: SB(bf16_t, float, bf16_t); break;
        case 6: SB(bf16_t, bf16_t, float); break;
        default: SB(bf16_t, bf16_t, bf16_t); break;
    }
#undef SB
    STDIFF_LAUNCH_STATUS();
    return STDIFF_OK;
}

extern "C" int stdiff_colsum(const void* dY, int ld, int dtype, float* db, int accumulate, int R, int C, void* stream) {
    STDIFF_CHECK_ARG(dY && db && R >= 0 && C > 0 && ld >= C);
    STDIFF_CHECK_ARG(dtype == STDIFF_F32 || dtype == STDIFF_BF16);
    dim3 grid(cdiv(C, 64)), block(256);
    const bool v4 = C % 4 == 0 && ld % 4 == 0 && ((uintptr_t)db & 15) == 0 &&
                    ((uintptr_t)dY & (dtype == STDIFF_BF16 ? 7 : 15)) == 0;
    if (v4) {
        dim3 g4(cdiv(C, 32));
        if (dtype == STDIFF_BF16)
            hipLaunchKernelGGL(colsum4_kernel<bf16_t>, g4, block, 0, (hipStream_t)stream, (const bf16_t*)dY, ld, db,
                               accumulate, R, C);
        else
            hipLaunchKernelGGL(colsum4_kernel<float>, g4, block, 0, (hipStream_t)stream, (const float*)dY, ld, db,
                               accumulate, R, C);
        STDIFF_LAUNCH_STATUS();
        return STDIFF_OK;
    }
    if (dtype == STDIFF_BF16)
        hipLaunchKernelGGL(colsum_kernel<bf16_t>, grid, block, 0, (hipStream_t)stream, (const bf16_t*)dY, ld, db,
                           accumulate, R, C);
    else
        hipLaunchKernelGGL(colsum_kernel<float>, grid, block, 0, (hipStream_t)stream, (const float*)dY, ld, db,
                           accumulate, R, C);
    STDIFF_LAUNCH_STATUS();
    return STDIFF_OK;
}

extern "C" int stdiff_grad_sumsq(const float* grads, int64_t n, float* sumsq, void* stream) {
    STDIFF_CHECK_ARG(grads && sumsq && n >= 0 && al16(grads));
    if (n == 0) return STDIFF_OK;
    // 2 blocks per CU: the per-bucket calls of the grouped backward run beside the latency-bound dX chain (sweep on the
    // training iteration, H = 1024 / configs[2] shape: 1024 blocks 2.46 / 6.12 ms, 512: 2.38 / 6.08, 256: 2.37 / 6.24, 128:
    // 2.46 / 6.78)
    static const int sblocks = [] { const char* e = getenv("STDIFF_SUMSQ_BLOCKS"); return e && atoi(e) > 0 ? atoi(e) : 2 * cu_count(); }();
    const unsigned gridx = (unsigned)std::min<size_t>(((size_t)n / 4 + 256) / 256, (size_t)sblocks);
    hipLaunchKernelGGL(sumsq_kernel, dim3(gridx), dim3(256), 0, (hipStream_t)stream, grads, n, sumsq);
    STDIFF_LAUNCH_STATUS();
    return STDIFF_OK;
}

extern "C" int stdiff_sum_parts(const float* parts, long long n, float* out, int accumulate, void* stream) {
    STDIFF_CHECK_ARG(out && n >= 0 && (parts || n == 0));
    hipLaunchKernelGGL(sum_parts_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, parts, (int64_t)n, out, accumulate);
    STDIFF_LAUNCH_STATUS();
    return STDIFF_OK;
}

namespace {
int adamw_check(const stdiff_adamw_args* a) {
    STDIFF_CHECK_ARG(a && a->params && a->grads && a->exp_avg && a->exp_avg_sq && a->sumsq && a->step_dev && a->scal &&
                     a->n >= 0);
    STDIFF_CHECK_ARG(al16(a->params) && al16(a->grads) && al16(a->exp_avg) && al16(a->exp_avg_sq) && al8(a->bf16_mirror));
    return STDIFF_OK;
}
}  // namespace

// The two halves of stdiff_clip_adamw for callers that update the flat buffer range by range (one range per layer, so
// that the next forward can start on a layer as soon as ITS parameters are updated): the scalars once, then any number
// of ranges, each the same arithmetic on elements [lo, lo + count).
extern "C" int stdiff_adamw_prepare(const stdiff_adamw_args* a, void* stream) {
    int rc = adamw_check(a);
    if (rc != STDIFF_OK) return rc;
    hipLaunchKernelGGL(adam_prep_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a->sumsq, a->step_dev, a->scal,
                       a->max_norm, a->grad_scale, a->lr, a->beta1, a->beta2);
    STDIFF_LAUNCH_STATUS();
    return STDIFF_OK;
}

extern "C" int stdiff_adamw_range(const stdiff_adamw_args* a, int64_t lo, int64_t count, void* stream) {
    int rc = adamw_check(a);
    if (rc != STDIFF_OK) return rc;
    STDIFF_CHECK_ARG(lo >= 0 && count >= 0 && lo + count <= a->n);
    if (lo % 4) return STDIFF_ERR_ALIGN;
    if (count == 0) return STDIFF_OK;
    // ONE block per CU: a range runs beside the next forward's batch-sized GEMMs, which are latency-bound -- with two
    // blocks per CU (the stand-alone optimum) the streaming update saturates the memory system and those GEMMs run 2-4x
    // slower.  Measured on the overlapped iteration (STDIFF_ADAMW_RANGE_BLOCKS sweep, H = 1024 / configs[2] shape):
    // 192: 2.47 / 6.69 ms, 256: 2.44 / 6.31, 320: 2.64 / 6.86, 512: 2.60 / 6.64.
    static const int xblocks = [] { const char* e = getenv("STDIFF_ADAMW_RANGE_BLOCKS"); return e && atoi(e) > 0 ? atoi(e) : cu_count(); }();
    dim3 grid((unsigned)std::min<size_t>(((size_t)count / 4 + 256) / 256, (size_t)xblocks));
    hipLaunchKernelGGL(clip_adamw_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, a->params + lo, a->grads + lo,
                       a->exp_avg + lo, a->exp_avg_sq + lo, count, a->scal, a->lr, a->beta1, a->beta2, a->eps,
                       a->weight_decay, a->bf16_mirror ? (bf16_t*)a->bf16_mirror + lo : nullptr);
    STDIFF_LAUNCH_STATUS();
    return STDIFF_OK;
}

extern "C" int stdiff_clip_adamw(const stdiff_adamw_args* a, void* stream) {
    int rc0 = adamw_check(a);
    if (rc0 != STDIFF_OK) return rc0;
    if (a->n == 0) return STDIFF_OK;
    hipLaunchKernelGGL(adam_prep_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a->sumsq, a->step_dev, a->scal,
                       a->max_norm, a->grad_scale, a->lr, a->beta1, a->beta2);
    // Grid: 2 blocks per CU walking the four arrays grid-strided, streaming (nontemporal) accesses.  Measured at 120 M
    // parameters (STDIFF_ADAMW_BLOCKS x STDIFF_ADAMW_NT sweep): 2048 blocks, cached: 780 us (4.6 TB/s); 512 blocks,
    // nontemporal: 607 us (5.9 TB/s); multiples of the CU count are the good grids (384: 732 us, 768: 652 us).
    static const int xblocks = [] { const char* e = getenv("STDIFF_ADAMW_BLOCKS"); return e && atoi(e) > 0 ? atoi(e) : 2 * cu_count(); }();
    static const int xnt = [] { const char* e = getenv("STDIFF_ADAMW_NT"); return e ? atoi(e) : 1; }();
    dim3 grid((unsigned)std::min<size_t>(((size_t)a->n / 4 + 256) / 256, (size_t)xblocks));
    if (xnt)
        hipLaunchKernelGGL(clip_adamw_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream,
                           a->params, a->grads, a->exp_avg, a->exp_avg_sq, a->n, a->scal, a->lr, a->beta1, a->beta2,
                           a->eps, a->weight_decay, (bf16_t*)a->bf16_mirror);
    else
        hipLaunchKernelGGL(clip_adamw_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream,
                           a->params, a->grads, a->exp_avg, a->exp_avg_sq, a->n, a->scal, a->lr, a->beta1, a->beta2,
                           a->eps, a->weight_decay, (bf16_t*)a->bf16_mirror);
    STDIFF_LAUNCH_STATUS();
    return STDIFF_OK;
}
