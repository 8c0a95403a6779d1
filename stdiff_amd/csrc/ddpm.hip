// Diffusion arithmetic, all HBM-bound and fused to one pass each:
//   reverse step + known-gene splice   (stDiff_scheduler.py:68-139, model/sample.py:94-95)
//   x_T initialisation                 (model/sample.py:56-60)
//   forward add_noise + train splice   (stDiff_scheduler.py:141-148, stDiff_train.py:65-70)
//   masked MSE forward + its gradient  (stDiff_train.py:75)
// The fp32 arithmetic uses explicit round-to-nearest mul/add in the reference's order (no FMA contraction), so
// with injected noise the results are bit-identical to the reference's eager ops.
// Each thread owns 4 consecutive genes of one row (16-byte accesses when G % 4 == 0).
#include "common.h"

// HIP compiles with -ffp-contract=fast and __fmul_rn/__fadd_rn are plain operators in its headers, so the
// a*b+c chains below would be fused into v_fma_f32 and drift 1 ulp from the reference's separate eager ops.
#pragma clang fp contract(off)

namespace {

// defined AFTER the pragma: these operations carry no 'contract' flag, so they survive inlining unfused
__device__ __forceinline__ float mulf(float a, float b) { return a * b; }
__device__ __forceinline__ float addf(float a, float b) { return a + b; }
__device__ __forceinline__ float subf(float a, float b) { return a - b; }

// ---------------------------------------------------------------- Philox4x32-10 + Box-Muller
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        uint32_t h0 = __umulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
        uint32_t h1 = __umulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
        uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0; c1 = l1; c2 = n2; c3 = l0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// 4 standard normals for (global row, column group, stream tag); independent of grid shape and world size
__device__ __forceinline__ f32x4 normal4(uint64_t seed, int64_t grow, uint32_t cgroup, uint32_t tag) {
    uint32_t r[4];
    philox4x32_10((uint32_t)grow, (uint32_t)((uint64_t)grow >> 32), cgroup, tag, (uint32_t)seed,
                  (uint32_t)(seed >> 32), r);
    f32x4 z;
    // Box-Muller on the hardware transcendentals: v_log_f32 is log2, v_sin_f32 / v_cos_f32 take their argument in
    // REVOLUTIONS (sin(2 pi u) = v_sin(u)), so no range reduction is needed for u in (0, 1).  ~1e-6 absolute error:
    // noise quality, not parity (parity runs inject their noise); the libm calls were a third of the step kernel's time
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        float u1 = ((float)(r[2 * i] >> 8) + 0.5f) * (1.0f / 16777216.0f);   // (0,1)
        float u2 = ((float)(r[2 * i + 1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
        float rad = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));      // sqrt(-2 ln u1)
        z[2 * i] = rad * __builtin_amdgcn_cosf(u2);
        z[2 * i + 1] = rad * __builtin_amdgcn_sinf(u2);
    }
    return z;
}

template <bool VEC> __device__ __forceinline__ f32x4 load4(const float* p, int n) {
    if constexpr (VEC) return *(const f32x4*)p;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < n; ++j) v[j] = p[j];
    return v;
}
template <bool VEC> __device__ __forceinline__ void store4(float* p, f32x4 v, int n) {
    if constexpr (VEC) { *(f32x4*)p = v; return; }
    for (int j = 0; j < n; ++j) p[j] = v[j];
}
template <typename T> __device__ __forceinline__ void store4p(T* p, f32x4 v, int n, bool vec) {
    if (vec) { st4(p, v); return; }
    for (int j = 0; j < n; ++j) Elem<T>::st(p + j, v[j]);
}

__device__ __forceinline__ float splice(float v, float m, float g) {
    // x*(1-mask) + mask*gt  (model/sample.py:95)
    return addf(mulf(v, subf(1.0f, m)), mulf(m, g));
}

// Rows are walked by the grid, column groups (4 genes) by the threads of a block: no 64-bit division per element, and
// a thread's mask quad is loaded once.  With in-kernel noise (z_base == NULL: the throughput mode) a column
// group whose four genes are all KNOWN (mask == 1) needs neither x, eps nor noise: x*(1-1) + 1*gt == gt for every finite
// update, so the group is a copy of gt (10 of 18 bytes per element, no Philox / Box-Muller; 41 % of the groups at 20 %
// held-out genes).  With injected noise (parity runs) every element goes through the reference's arithmetic.
template <bool VEC, typename TP>
__global__ __launch_bounds__(256) void ddpm_step_kernel(stdiff_step_args a, int cols_t) {
    // the block's 256 threads form cols_t (a power of two) column lanes x 256 / cols_t rows
    const int tx = threadIdx.x & (cols_t - 1), ty = threadIdx.x / cols_t, rpb = blockDim.x / cols_t;
    const int t = *a.t_dev;
    const float ca = a.coef[t * 5 + 0], cb = a.coef[t * 5 + 1], c1 = a.coef[t * 5 + 2], c2 = a.coef[t * 5 + 3],
                sg = a.coef[t * 5 + 4];
    const int gq = (a.G + 3) >> 2;
    const float* zb = a.z_base ? a.z_base + (size_t)(a.t_first - t) * (size_t)(a.z_rows > 0 ? a.z_rows : a.R) * a.G : nullptr;
    const bool pvec = VEC && (a.ldxp % 4 == 0);
    for (int cg = tx; cg < gq; cg += cols_t) {
        const int c = cg * 4, n = min(4, a.G - c);
        const f32x4 m = load4<VEC>(a.mask + c, n);
        const bool known = !zb && m[0] == 1.0f && (n < 2 || m[1] == 1.0f) && (n < 3 || m[2] == 1.0f) && (n < 4 || m[3] == 1.0f);
        // 4 rows per pass, every load of the pass issued before the first use (memory-level parallelism per thread)
        constexpr int U = 4;
        const int rstep = gridDim.x * rpb;
        for (int r0 = blockIdx.x * rpb + ty; r0 < a.R; r0 += U * rstep) {
            f32x4 g[U], x[U], e[U], z[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int r = r0 + u * rstep;
                if (r < a.R) {
                    const size_t off = (size_t)r * a.G + c;
                    g[u] = load4<VEC>(a.gt + off, n);
                    if (!known) {
                        x[u] = load4<VEC>(a.x + off, n);
                        e[u] = load4<VEC>(a.eps + off, n);
                        if (t > 0 && zb) z[u] = load4<VEC>(zb + off, n);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int r = r0 + u * rstep;
                if (r >= a.R) break;
                const size_t off = (size_t)r * a.G + c;
                f32x4 o = g[u];
                if (!known) {
                    f32x4 zz = {0.f, 0.f, 0.f, 0.f};
                    if (t > 0) zz = zb ? z[u] : normal4(a.seed, a.row_offset + r, (uint32_t)cg, (uint32_t)t);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float x0 = a.pred_type == 0
                                       ? fminf(fmaxf(subf(mulf(ca, x[u][j]), mulf(cb, e[u][j])), -1.0f), 1.0f)
                                       : e[u][j];
                        float mu = addf(mulf(c1, x0), mulf(c2, x[u][j]));
                        if (t > 0) mu = addf(mu, mulf(sg, zz[j]));
                        o[j] = splice(mu, m[j], g[u][j]);
                    }
                }
                store4<VEC>(a.x + off, o, n);
                if (a.xp) store4p((TP*)a.xp + (size_t)r * a.ldxp + c, o, n, pvec);
            }
        }
    }
}

__global__ void advance_kernel(int32_t* t) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *t -= 1;
}

// every block waits until the 100 MHz realtime counter has advanced by `ticks` (an exit condition every wave reaches)
__global__ __launch_bounds__(64) void spin_kernel(unsigned long long ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

template <bool VEC, typename TP>
__global__ __launch_bounds__(256) void init_sample_kernel(float* x, const float* zin, uint64_t seed, int64_t row_offset,
                                                          const float* gt, const float* mask, TP* xp, int ldxp, int R,
                                                          int G) {
    const int gq = (G + 3) >> 2;
    const size_t total = (size_t)R * gq;
    const bool pvec = VEC && (ldxp % 4 == 0);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / gq), cg = (int)(i % gq), c = cg * 4;
        const int n = min(4, G - c);
        const size_t off = (size_t)r * G + c;
        f32x4 z = zin ? load4<VEC>(zin + off, n) : normal4(seed, row_offset + r, (uint32_t)cg, 0xFFFFFFFFu);
        f32x4 g = load4<VEC>(gt + off, n), m = load4<VEC>(mask + c, n), o;
#pragma unroll
        for (int j = 0; j < 4; ++j)  // x_t*(1-mask) + gt*mask  (model/sample.py:60)
            o[j] = addf(mulf(z[j], subf(1.0f, m[j])), mulf(g[j], m[j]));
        store4<VEC>(x + off, o, n);
        if (xp) store4p(xp + (size_t)r * ldxp + c, o, n, pvec);
    }
}

template <bool VEC, typename TP>
__global__ __launch_bounds__(256) void add_noise_splice_kernel(const float* x, const float* noise, const int64_t* t,
                                                               const float* sa, const float* sb, const float* mask,
                                                               float* xn, TP* xp, int ldxp, int R, int G) {
    const int gq = (G + 3) >> 2;
    const size_t total = (size_t)R * gq;
    const bool pvec = VEC && (ldxp % 4 == 0);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / gq), c = (int)(i % gq) * 4;
        const int n = min(4, G - c);
        const size_t off = (size_t)r * G + c;
        const float s1 = sa[t[r]], s2 = sb[t[r]];
        f32x4 xv = load4<VEC>(x + off, n), e = load4<VEC>(noise + off, n), m = load4<VEC>(mask + c, n), o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float xt = addf(mulf(s1, xv[j]), mulf(s2, e[j]));                       // scheduler :148
            o[j] = addf(mulf(xt, subf(1.0f, m[j])), mulf(xv[j], m[j]));        // train :70
        }
        if (xn) store4<VEC>(xn + off, o, n);
        if (xp) store4p(xp + (size_t)r * ldxp + c, o, n, pvec);
    }
}

template <bool VEC, typename TP>
__global__ __launch_bounds__(256) void masked_mse_kernel(const float* pred, const float* noise, const float* mask,
                                                         float* loss, float* dpred, TP* dp, int lddp, float gscale,
                                                         int R, int G) {
    const int gq = (G + 3) >> 2;
    const size_t total = (size_t)R * gq;
    const float inv = 1.0f / ((float)R * (float)G);
    const bool pvec = VEC && (lddp % 4 == 0);
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / gq), c = (int)(i % gq) * 4;
        const int n = min(4, G - c);
        const size_t off = (size_t)r * G + c;
        f32x4 p = load4<VEC>(pred + off, n), e = load4<VEC>(noise + off, n), m = load4<VEC>(mask + c, n), d;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float w = 1.0f - m[j];
            float diff = e[j] * w - p[j] * w;      // criterion(noise*(1-mask), noise_pred*(1-mask))
            if (j < n) acc += diff * diff;
            d[j] = -2.0f * diff * w * inv * gscale;
        }
        if (dpred) store4<VEC>(dpred + off, d, n);
        if (dp) store4p(dp + (size_t)r * lddp + c, d, n, pvec);
    }
    acc = wave_sum(acc);
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss, (part[0] + part[1] + part[2] + part[3]) * inv);
}

inline int ew_grid(size_t n) {
    // grid-stride cap, swept 512 ... 16384 at 20 000 x 1000 (ddpm_step: 59.4 / 59.6 / 59.6 / 57.3 / 54.5 us)
    static const int cap = [] { const char* e = getenv("STDIFF_EW_BLOCKS"); return e && atoi(e) > 0 ? atoi(e) : 256 * 64; }();
    return (int)std::min<size_t>((n + 255) / 256, (size_t)cap);
}
inline bool al16(const void* p) { return p == nullptr || ((uintptr_t)p & 15) == 0; }

}  // namespace

#define DISPATCH_VEC_TP(KERNEL, vec, bf, grid, st, ...)                                                   \
    do {                                                                                                  \
        if (vec) {                                                                                        \
            if (bf) hipLaunchKernelGGL((KERNEL<true, bf16_t>), grid, dim3(256), 0, st, __VA_ARGS__);      \
            else hipLaunchKernelGGL((KERNEL<true, float>), grid, dim3(256), 0, st, __VA_ARGS__);          \
        } else {                                                                                          \
            if (bf) hipLaunchKernelGGL((KERNEL<false, bf16_t>), grid, dim3(256), 0, st, __VA_ARGS__);     \
            else hipLaunchKernelGGL((KERNEL<false, float>), grid, dim3(256), 0, st, __VA_ARGS__);         \
        }                                                                                                 \
    } while (0)

extern "C" int stdiff_ddpm_step(const stdiff_step_args* a, void* stream) {
    STDIFF_CHECK_ARG(a && a->x && a->eps && a->gt && a->mask && a->coef && a->t_dev);
    STDIFF_CHECK_ARG(a->R >= 0 && a->G > 0 && (a->pred_type == 0 || a->pred_type == 1));
    STDIFF_CHECK_ARG(!a->xp || a->ldxp >= a->G);
    STDIFF_CHECK_ARG(a->z_rows == 0 || a->z_rows >= a->R);
    if (a->R == 0) return STDIFF_OK;
    const bool vec = a->G % 4 == 0 && al16(a->x) && al16(a->eps) && al16(a->gt) && al16(a->mask) && al16(a->z_base) &&
                     al16(a->xp);
    const int gq = (a->G + 3) / 4;
    int cols_t = 1;
    while (cols_t < gq && cols_t < 256) cols_t <<= 1;
    const int rpb = 256 / cols_t;
    dim3 grid(std::min<size_t>(((size_t)a->R + rpb - 1) / rpb, (size_t)ew_grid((size_t)a->R * gq)));
    DISPATCH_VEC_TP(ddpm_step_kernel, vec, a->xp_dtype == STDIFF_BF16, grid, (hipStream_t)stream, *a, cols_t);
    STDIFF_LAUNCH_STATUS();
    return STDIFF_OK;
}

extern "C" int stdiff_advance_timestep(int32_t* t_dev, void* stream) {
    STDIFF_CHECK_ARG(t_dev);
    hipLaunchKernelGGL(advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, t_dev);
    STDIFF_LAUNCH_STATUS();
    return STDIFF_OK;
}

extern "C" int stdiff_spin_probe(int blocks, int microseconds, void* stream) {
    STDIFF_CHECK_ARG(blocks > 0 && blocks <= 4096 && microseconds > 0 && microseconds <= 100000);
    hipLaunchKernelGGL(spin_kernel, dim3(blocks), dim3(64), 0, (hipStream_t)stream, 100ull * microseconds);
    STDIFF_LAUNCH_STATUS();
    return STDIFF_OK;
}

extern "C" int stdiff_init_sample(float* x, const float* z, uint64_t seed, int64_t row_offset, const float* gt,
                                  const float* mask, void* xp, int ldxp, int xp_dtype, int R, int G, void* stream) {
    STDIFF_CHECK_ARG(x && gt && mask && R >= 0 && G > 0 && (!xp || ldxp >= G));
    if (R == 0) return STDIFF_OK;
    const bool vec = G % 4 == 0 && al16(x) && al16(z) && al16(gt) && al16(mask) && al16(xp);
    dim3 grid(ew_grid((size_t)R * ((G + 3) / 4)));
    if (xp_dtype == STDIFF_BF16) {
        if (vec) hipLaunchKernelGGL((init_sample_kernel<true, bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, x, z, seed, row_offset, gt, mask, (bf16_t*)xp, ldxp, R, G);
        else hipLaunchKernelGGL((init_sample_kernel<false, bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, x, z, seed, row_offset, gt, mask, (bf16_t*)xp, ldxp, R, G);
    } else {
        if (vec) hipLaunchKernelGGL((init_sample_kernel<true, float>), grid, dim3(256), 0, (hipStream_t)stream, x, z, seed, row_offset, gt, mask, (float*)xp, ldxp, R, G);
        else hipLaunchKernelGGL((init_sample_kernel<false, float>), grid, dim3(256), 0, (hipStream_t)stream, x, z, seed, row_offset, gt, mask, (float*)xp, ldxp, R, G);
    }
    STDIFF_LAUNCH_STATUS();
    return STDIFF_OK;
}

extern "C" int stdiff_add_noise_splice(const float* x, const float* noise, const int64_t* t, const float* sqrt_ac,
                                       const float* sqrt_1mac, const float* mask, float* x_noisy, void* xp, int ldxp,
                                       int xp_dtype, int R, int G, void* stream) {
    STDIFF_CHECK_ARG(x && noise && t && sqrt_ac && sqrt_1mac && mask && (x_noisy || xp) && R >= 0 && G > 0);
    STDIFF_CHECK_ARG(!xp || ldxp >= G);
    if (R == 0) return STDIFF_OK;
    const bool vec = G % 4 == 0 && al16(x) && al16(noise) && al16(mask) && al16(x_noisy) && al16(xp);
    dim3 grid(ew_grid((size_t)R * ((G + 3) / 4)));
    if (xp_dtype == STDIFF_BF16) {
        if (vec) hipLaunchKernelGGL((add_noise_splice_kernel<true, bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, x, noise, t, sqrt_ac, sqrt_1mac, mask, x_noisy, (bf16_t*)xp, ldxp, R, G);
        else hipLaunchKernelGGL((add_noise_splice_kernel<false, bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, x, noise, t, sqrt_ac, sqrt_1mac, mask, x_noisy, (bf16_t*)xp, ldxp, R, G);
    } else {
        if (vec) hipLaunchKernelGGL((add_noise_splice_kernel<true, float>), grid, dim3(256), 0, (hipStream_t)stream, x, noise, t, sqrt_ac, sqrt_1mac, mask, x_noisy, (float*)xp, ldxp, R, G);
        else hipLaunchKernelGGL((add_noise_splice_kernel<false, float>), grid, dim3(256), 0, (hipStream_t)stream, x, noise, t, sqrt_ac, sqrt_1mac, mask, x_noisy, (float*)xp, ldxp, R, G);
    }
    STDIFF_LAUNCH_STATUS();
    return STDIFF_OK;
}

extern "C" int stdiff_masked_mse(const float* pred, const float* noise, const float* mask, float* loss, float* dpred,
                                 void* dp, int lddp, int dp_dtype, float grad_scale, int R, int G, void* stream) {
    STDIFF_CHECK_ARG(pred && noise && mask && loss && R > 0 && G > 0 && (!dp || lddp >= G));
    const bool vec = G % 4 == 0 && al16(pred) && al16(noise) && al16(mask) && al16(dpred) && al16(dp);
    dim3 grid(ew_grid((size_t)R * ((G + 3) / 4)));
    if (dp_dtype == STDIFF_BF16) {
        if (vec) hipLaunchKernelGGL((masked_mse_kernel<true, bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, pred, noise, mask, loss, dpred, (bf16_t*)dp, lddp, grad_scale, R, G);
        else hipLaunchKernelGGL((masked_mse_kernel<false, bf16_t>), grid, dim3(256), 0, (hipStream_t)stream, pred, noise, mask, loss, dpred, (bf16_t*)dp, lddp, grad_scale, R, G);
    } else {
        if (vec) hipLaunchKernelGGL((masked_mse_kernel<true, float>), grid, dim3(256), 0, (hipStream_t)stream, pred, noise, mask, loss, dpred, (float*)dp, lddp, grad_scale, R, G);
        else hipLaunchKernelGGL((masked_mse_kernel<false, float>), grid, dim3(256), 0, (hipStream_t)stream, pred, noise, mask, loss, dpred, (float*)dp, lddp, grad_scale, R, G);
    }
    STDIFF_LAUNCH_STATUS();
    return STDIFF_OK;
}
