// Shared device helpers for the stDiff CDNA4 (gfx950) kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/stdiff_hip.h"

// The 16-bit operand format is a property of the BUILD: libstdiff_hip.so computes in bf16 (8-bit mantissa, fp32
// range), libstdiff_hip_f16.so (-DSTDIFF_H16_FP16, sampling only) in IEEE fp16 (11-bit mantissa: per-call error of the
// denoiser 8x smaller at the same MFMA rate; conversions saturate at +-65504).  Both are dtype code STDIFF_BF16 (= "the
// library's 16-bit format") at the C ABI; `bf16_t` / `bf16x8` below name the 16-bit storage and MFMA fragment types of
// either build.
#ifdef STDIFF_H16_FP16
typedef _Float16 h16;
#define STDIFF_MFMA_16x16x32_H16 __builtin_amdgcn_mfma_f32_16x16x32_f16
#define STDIFF_MFMA_32x32x16_H16 __builtin_amdgcn_mfma_f32_32x32x16_f16
typedef __fp16 tr16_raw __attribute__((__vector_size__(4 * sizeof(__fp16))));
#define STDIFF_TR16_BUILTIN __builtin_amdgcn_ds_read_tr16_b64_v4f16
#else
typedef __bf16 h16;
#define STDIFF_MFMA_16x16x32_H16 __builtin_amdgcn_mfma_f32_16x16x32_bf16
#define STDIFF_MFMA_32x32x16_H16 __builtin_amdgcn_mfma_f32_32x32x16_bf16
typedef __bf16 tr16_raw __attribute__((__vector_size__(4 * sizeof(__bf16))));
#define STDIFF_TR16_BUILTIN __builtin_amdgcn_ds_read_tr16_b64_v4bf16
#endif
typedef unsigned short bf16_t;  // storage type of a 16-bit element
typedef __attribute__((ext_vector_type(8))) h16 bf16x8;
typedef __attribute__((ext_vector_type(4))) h16 bf16x4;
// ds_read_b64_tr_b16: the hardware transposes a 4-row x 16-column block of 16-bit elements per 16 lanes
__device__ __forceinline__ bf16x4 ds_read_tr16_b64(const void* lds_ptr) {
    return __builtin_bit_cast(bf16x4, STDIFF_TR16_BUILTIN((__attribute__((address_space(3))) tr16_raw*)lds_ptr));
}
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define WAVE 64

// Status codes follow include/stdiff_hip.h: 0 ok, <0 bad argument, >0 hipError_t.
#define STDIFF_CHECK_ARG(cond) \
    do {                       \
        if (!(cond)) return STDIFF_ERR_ARG; \
    } while (0)
#define STDIFF_LAUNCH_STATUS()                 \
    do {                                       \
        hipError_t e__ = hipGetLastError();    \
        if (e__ != hipSuccess) return (int)e__; \
    } while (0)

typedef float f32x2_cv __attribute__((ext_vector_type(2)));
typedef h16 bf16x2_cv __attribute__((ext_vector_type(2)));
#ifdef STDIFF_H16_FP16
#ifdef STDIFF_H16_NOSAT   // A/B timing only: conversions overflow to inf
__device__ __forceinline__ float h16_sat(float f) { return f; }
#else
__device__ __forceinline__ float h16_sat(float f) { return __builtin_amdgcn_fmed3f(f, -65504.0f, 65504.0f); }
#endif
__device__ __forceinline__ float bf2f(bf16_t u) { return (float)__builtin_bit_cast(_Float16, u); }
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (_Float16)h16_sat(f)); }  // RNE
__device__ __forceinline__ float bf_lo(unsigned u) { return bf2f((bf16_t)(u & 0xffffu)); }
__device__ __forceinline__ float bf_hi(unsigned u) { return bf2f((bf16_t)(u >> 16)); }
__device__ __forceinline__ unsigned pack2bf(float a, float b) {
    f32x2_cv v = {h16_sat(a), h16_sat(b)};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_cv));
}
#else
__device__ __forceinline__ float bf2f(bf16_t u) { return __builtin_bit_cast(float, (unsigned)u << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }  // RNE, NaN-safe
__device__ __forceinline__ float bf_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
// two floats -> packed bf16 pair (RNE), one v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned pack2bf(float a, float b) {
    f32x2_cv v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_cv));
}
#endif

template <typename T> struct Elem;
template <> struct Elem<float> {
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
    static __device__ __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
    static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
};

// 4 consecutive elements <-> 4 floats (16-B access for f32, 8-B for bf16). p must be aligned to that.
__device__ __forceinline__ f32x4 ld4(const float* p) { return *(const f32x4*)p; }
__device__ __forceinline__ f32x4 ld4(const bf16_t* p) {
    u32x2 r = *(const u32x2*)p;
    f32x4 v;
    v[0] = bf_lo(r[0]);
    v[1] = bf_hi(r[0]);
    v[2] = bf_lo(r[1]);
    v[3] = bf_hi(r[1]);
    return v;
}
__device__ __forceinline__ void st4(float* p, f32x4 v) { *(f32x4*)p = v; }
__device__ __forceinline__ void st4(bf16_t* p, f32x4 v) {
    u32x2 r;
    r[0] = pack2bf(v[0], v[1]);
    r[1] = pack2bf(v[2], v[3]);
    *(u32x2*)p = r;
}

__device__ __forceinline__ float silu_f(float v) { return v / (1.0f + expf(-v)); }
__device__ __forceinline__ float gelu_erf_f(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f)); }

// GELU(erf) for bf16 OUTPUTS, 4 lanes at a time, no transcendentals (packs into v_pk_fma_f32): Phi(x) = 0.5 + xc*Q(u),
// xc = clamp(x, -5, 5), u = 2*xc^2/25 - 1, Q = degree-12 minimax fit (tools/fit_gelu.py).  |Phi - exact| <= 2.6e-7,
// |gelu - exact| <= 1.3e-6 on [-5, 5] and <= 1.5e-6*|x|/5 beyond: 3 orders below a bf16 ulp of the result scale.
__device__ __forceinline__ f32x4 gelu_phi4(f32x4 x) {
    constexpr float c[13] = {1.413637847e-01f, -7.029628009e-02f, 5.152052268e-02f, -4.044260830e-02f,
                             3.144217283e-02f, -2.326865308e-02f, 1.640217379e-02f, -1.116433926e-02f,
                             6.410510279e-03f, -2.685575048e-03f, 1.722799498e-03f, -1.613265136e-03f,
                             6.087272777e-04f};
    f32x4 xc;
#pragma unroll
    for (int j = 0; j < 4; ++j) xc[j] = __builtin_amdgcn_fmed3f(x[j], -5.0f, 5.0f);
    const f32x4 u = xc * xc * 0.08f - 1.0f;
    f32x4 q = u * c[12] + c[11];
#pragma unroll
    for (int k = 10; k >= 0; --k) q = q * u + c[k];
    return xc * q + 0.5f;
}
__device__ __forceinline__ f32x4 gelu_poly4(f32x4 x) { return x * gelu_phi4(x); }
// d gelu / dx = Phi(x) + x * phi(x) for 16-bit OUTPUTS: Phi from the polynomial above, the density through the
// hardware exp2 (relative error ~1e-7 of a value <= 0.4): far below the output rounding
__device__ __forceinline__ f32x4 gelu_grad4(f32x4 x) {
    f32x4 d = gelu_phi4(x);
#pragma unroll
    for (int j = 0; j < 4; ++j)
        d[j] += x[j] * (0.3989422804014327f * __builtin_amdgcn_exp2f(-0.72134752044448170368f * x[j] * x[j]));
    return d;
}

// LayerNorm + modulate of one element, written once so that every kernel that applies it (ln_modulate_kernel and the
// LN-modulate epilogue of the adaLN GEMM) rounds identically:  ((v - mean) * rstd) * (1 + scale) + shift, the last
// multiply-add as ONE fused operation.
__device__ __forceinline__ float ln_mod_apply(float v, float mean, float rstd, float scale, float shift) {
    const float t = (v - mean) * rstd;
    return __builtin_fmaf(t, 1.0f + scale, shift);
}

// the same with (1 + scale) formed beforehand (the GEMM epilogue prepares it while its loads are in flight): same bits
__device__ __forceinline__ float ln_mod_apply_pre(float v, float mean, float rstd, float one_plus_scale, float shift) {
    const float t = (v - mean) * rstd;
    return __builtin_fmaf(t, one_plus_scale, shift);
}

// Sum of split-K partial products in slice order (the order of splitk_reduce_kernel), with every slice's load issued
// before the first add: the consumers that call this run one wave per row at two waves per CU, where a chain of
// dependent loads would cost a memory round trip per slice.  slices <= STDIFF_MAX_SLICES (the host's split limit).
#define STDIFF_MAX_SLICES 16
__device__ __forceinline__ f32x4 sum_slices(const float* w, int slices, long long stride) {
    f32x4 part[STDIFF_MAX_SLICES];
#pragma unroll
    for (int k = 0; k < STDIFF_MAX_SLICES; ++k)
        if (k < slices) part[k] = *(const f32x4*)(w + (size_t)k * stride);
    f32x4 a = part[0];
#pragma unroll
    for (int k = 1; k < STDIFF_MAX_SLICES; ++k)
        if (k < slices) a += part[k];
    return a;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

static inline int cu_count() {      // CUs of the current device (one process drives one GPU model: cached once)
    static const int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        return v;
    }();
    return n;
}
static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
static inline size_t dtype_size(int dt) { return dt == STDIFF_BF16 ? 2 : 4; }
