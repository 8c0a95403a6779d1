// Cell-axis attention of Attention2.forward (stDiff_model.py:25-34), flash style on the gfx950 matrix cores.
//
// The "sequence" is the set of cells of one reference mini-batch (a chunk, <= batch_size rows): for every chunk and
// head, O = softmax(Q K^T d^-0.5) V over the cells of THAT chunk only (rows of different chunks never mix; the last
// chunk may be partial).  The [heads, C, C] score matrix the reference materialises never leaves registers.
//
//   grid  = (ceil(max_chunk/128), heads, n_chunks), block = 4 waves, wave = 32 query rows, K/V tile = 64 keys in LDS
//   S^T   = K . Q^T  with the KEY on the MFMA row and the QUERY on the lane (32x32 tiles): a lane then owns one
//           query's scores for 16 keys, the softmax row reduction is 15 in-register max/adds + ONE cross-half
//           exchange, and the fp32 P^T accumulator is already laid out as the B operand of  O^T += V^T . P^T
//           (cdna guide section 3, "An accumulator tile as the next MFMA's operand": element j of lane half h of
//           k-step s is key 16s + 8(j>>2) + 4h + (j&3)).
//   bf16  : v_mfma_f32_32x32x16_bf16; K fragments by ds_read_b128 from a [key][d] image padded by 16 B; V^T
//           fragments by ds_read_b64_tr_b16 (hardware transpose) from a [key][d] image whose row stride makes the
//           transposed reads conflict-free (tools/lds_conflicts.py).
//   f32   : v_mfma_f32_32x32x2_f32 (exact fp32), one fp32 register per operand.
#include "common.h"

namespace {

constexpr int BQ = 128, BKV = 64;

template <typename T, int D> struct ACfg;
template <int D> struct ACfg<bf16_t, D> {
    static constexpr int EPV = 8;                       // elements per 16-byte chunk
    static constexpr int KSTR = D * 2 + 16;             // K image row stride (bytes)
    static constexpr int VSTR = D == 32 ? 64 : (D == 64 ? 192 : 320);
};
template <int D> struct ACfg<float, D> {
    static constexpr int EPV = 4;
    static constexpr int KSTR = D * 4 + 16;
    static constexpr int VSTR = D * 4;
};

__device__ __forceinline__ int kappa(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

template <typename T, int D>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const T* __restrict__ qkv, int ldqkv, T* __restrict__ out,
                                                       int ldo, float* __restrict__ lse,
                                                       const int32_t* __restrict__ cu_rows, int heads, float c_log2) {
    using C = ACfg<T, D>;
    constexpr int EPV = C::EPV, KSTR = C::KSTR, VSTR = C::VSTR;
    constexpr int CPR = D / EPV;                 // 16-byte chunks per row
    constexpr int LPT = BKV * CPR / 256;         // chunks per thread per operand
    constexpr bool BF = sizeof(T) == 2;
    __shared__ __attribute__((aligned(16))) unsigned char kls[BKV * KSTR];
    __shared__ __attribute__((aligned(16))) unsigned char vls[BKV * VSTR];

    const int chunk = blockIdx.z, hd = blockIdx.y;
    const int r0 = cu_rows[chunk], len = cu_rows[chunk + 1] - r0;
    const int q0 = blockIdx.x * BQ;
    if (q0 >= len) return;
    const int H = heads * D;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int qi = lane & 31, hh = lane >> 5;
    const int qrow = q0 + wid * 32 + qi;
    const T* qp = qkv + (size_t)(r0 + min(qrow, len - 1)) * ldqkv + hd * D;

    // ---- Q fragments, kept in registers for the whole pass
    bf16x8 qb[BF ? D / 16 : 1];
    f32x4 qf[BF ? 1 : D / 8];
    if constexpr (BF) {
#pragma unroll
        for (int s = 0; s < D / 16; ++s) qb[s] = *(const bf16x8*)(qp + 16 * s + 8 * hh);
    } else {
#pragma unroll
        for (int u = 0; u < D / 8; ++u) qf[u] = *(const f32x4*)(qp + 8 * u + 4 * hh);
    }

    f32x16 o[D / 32];
#pragma unroll
    for (int b = 0; b < D / 32; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[b][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    // ---- K/V staging: thread -> (row, 16-byte chunk)
    const T* kg[LPT];
    int ks_off[LPT], vs_off[LPT], krow[LPT];
#pragma unroll
    for (int i = 0; i < LPT; ++i) {
        int c = tid + 256 * i;
        int row = c / CPR, ch = c % CPR;
        krow[i] = row;
        kg[i] = qkv + H + hd * D + ch * EPV;
        ks_off[i] = row * KSTR + ch * 16;
        vs_off[i] = row * VSTR + ch * 16;
    }
    u32x4 rk[LPT], rv[LPT];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const T* p = kg[i] + (size_t)(r0 + min(k0 + krow[i], len - 1)) * ldqkv;
            rk[i] = *(const u32x4*)p;
            rv[i] = *(const u32x4*)(p + H);
        }
    };
    auto sstore = [&]() {
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            *(u32x4*)(kls + ks_off[i]) = rk[i];
            *(u32x4*)(vls + vs_off[i]) = rv[i];
        }
    };

    const int ntiles = (len + BKV - 1) / BKV;
    if constexpr (BF) gload(0);
    for (int j = 0; j < ntiles; ++j) {
        const int k0 = j * BKV;
        __syncthreads();
        if constexpr (!BF) gload(k0);
        sstore();
        __syncthreads();
        if constexpr (BF) { if (j + 1 < ntiles) gload(k0 + BKV); }

#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            if (k0 + kt * 32 >= len) break;
            // ---- S^T[key, q] for 32 keys
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
            const unsigned char* kr = kls + (kt * 32 + qi) * KSTR;
            if constexpr (BF) {
#pragma unroll
                for (int ss = 0; ss < D / 16; ++ss) {
                    bf16x8 a = *(const bf16x8*)(kr + (16 * ss + 8 * hh) * 2);
                    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, qb[ss], s, 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int u = 0; u < D / 8; ++u) {
                    f32x4 a = *(const f32x4*)(kr + (8 * u + 4 * hh) * 4);
#pragma unroll
                    for (int i = 0; i < 4; ++i) s = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], qf[u][i], s, 0, 0, 0);
                }
            }
            // ---- online softmax in base 2; masked keys (past the chunk) get -inf
            const bool tail = k0 + kt * 32 + 32 > len;
            float mt = -INFINITY;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = s[r] * c_log2;
                if (tail && k0 + kt * 32 + kappa(r, hh) >= len) v = -INFINITY;
                s[r] = v;
                mt = fmaxf(mt, v);
            }
            mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
            const float m_new = fmaxf(m_run, mt);
            const float alpha = exp2f(m_run - m_new);
            float ps = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s[r] = exp2f(s[r] - m_new);
                ps += s[r];
            }
            l_run = l_run * alpha + ps;
            m_run = m_new;
#pragma unroll
            for (int b = 0; b < D / 32; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[b][r] *= alpha;
            // ---- O^T[dcol, q] += V^T . P^T
            if constexpr (BF) {
                bf16x8 pb[2];
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int e = 0; e < 8; ++e) pb[ks][e] = (__bf16)s[8 * ks + e];
                const int g1 = (lane >> 4) & 1, q4 = (lane & 15) >> 2, p4 = lane & 3;
#pragma unroll
                for (int b = 0; b < D / 32; ++b) {
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        const int kb = kt * 32 + 16 * ks + 4 * hh + q4;
                        const unsigned char* va = vls + kb * VSTR + (b * 32 + 16 * g1 + 4 * p4) * 2;
                        bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                            (__attribute__((address_space(3))) bf16x4*)(va));
                        bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
                            (__attribute__((address_space(3))) bf16x4*)(va + 8 * VSTR));
                        bf16x8 a = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                        o[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, pb[ks], o[b], 0, 0, 0);
                    }
                }
            } else {
#pragma unroll
                for (int b = 0; b < D / 32; ++b) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float a = *(const float*)(vls + (kt * 32 + kappa(r, hh)) * VSTR + (b * 32 + qi) * 4);
                        o[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, s[r], o[b], 0, 0, 0);
                    }
                }
            }
        }
    }

    // ---- normalise and store: lane (qi, hh) holds O[q][32b + 8g + 4hh + 0..3]
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;
    if (qrow < len) {
        T* op = out + (size_t)(r0 + qrow) * ldo + hd * D;
#pragma unroll
        for (int b = 0; b < D / 32; ++b)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v = {o[b][4 * g] * inv, o[b][4 * g + 1] * inv, o[b][4 * g + 2] * inv, o[b][4 * g + 3] * inv};
                st4(op + b * 32 + 8 * g + 4 * hh, v);
            }
        if (lse && hh == 0) lse[(size_t)(r0 + qrow) * heads + hd] = m_run + log2f(l_tot);
    }
}

template <typename T>
int launch_fwd(const void* qkv, int ldqkv, void* out, int ldo, float* lse, const int32_t* cu, int n_chunks, int maxc,
               int heads, int d, hipStream_t st) {
    dim3 grid(cdiv(maxc, BQ), heads, n_chunks), block(256);
    const float c = (1.0f / sqrtf((float)d)) * 1.4426950408889634f;
#define CASE(DD)                                                                                                 \
    hipLaunchKernelGGL((attn_fwd_kernel<T, DD>), grid, block, 0, st, (const T*)qkv, ldqkv, (T*)out, ldo, lse, cu, \
                       heads, c)
    if (d == 32) CASE(32);
    else if (d == 64) CASE(64);
    else if (d == 128) CASE(128);
    else return STDIFF_ERR_SHAPE;
#undef CASE
    STDIFF_LAUNCH_STATUS();
    return STDIFF_OK;
}

}  // namespace

extern "C" int stdiff_attention_fwd(const void* qkv, int ldqkv, void* out, int ldo, float* lse, const int32_t* cu_rows,
                                    int n_chunks, int max_chunk_rows, int heads, int head_dim, int dtype,
                                    void* stream) {
    STDIFF_CHECK_ARG(qkv && out && cu_rows && n_chunks >= 0 && max_chunk_rows >= 0 && heads > 0);
    STDIFF_CHECK_ARG(dtype == STDIFF_F32 || dtype == STDIFF_BF16);
    if (n_chunks == 0 || max_chunk_rows == 0) return STDIFF_OK;
    if (head_dim != 32 && head_dim != 64 && head_dim != 128) return STDIFF_ERR_SHAPE;
    if (heads > 65535 || n_chunks > 65535) return STDIFF_ERR_SHAPE;
    const int epv = dtype == STDIFF_BF16 ? 8 : 4;
    if (ldqkv % epv || ldo % 4 || ((uintptr_t)qkv & 15) || ((uintptr_t)out & 15)) return STDIFF_ERR_ALIGN;
    if (ldqkv < 3 * heads * head_dim || ldo < heads * head_dim) return STDIFF_ERR_ARG;
    if (dtype == STDIFF_BF16)
        return launch_fwd<bf16_t>(qkv, ldqkv, out, ldo, lse, cu_rows, n_chunks, max_chunk_rows, heads, head_dim,
                                  (hipStream_t)stream);
    return launch_fwd<float>(qkv, ldqkv, out, ldo, lse, cu_rows, n_chunks, max_chunk_rows, heads, head_dim,
                             (hipStream_t)stream);
}
