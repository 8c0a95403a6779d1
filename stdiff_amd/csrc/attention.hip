// Cell-axis attention of Attention2.forward (stDiff_model.py:25-34) and its backward, flash style on the gfx950
// matrix cores.
//
// The "sequence" is the set of cells of one reference mini-batch (a chunk, <= batch_size rows): for every chunk and
// head, O = softmax(Q K^T d^-0.5) V over the cells of THAT chunk only (rows of different chunks never mix; the last
// chunk may be partial).  The [heads, C, C] score matrix the reference materialises never leaves registers.
//
// Building blocks (a wave owns 32 "lane rows"; a 64-row tile of the other operand sits in LDS):
//   rows_dot_regs : X[tile row, lane row] = sum_d Tile[row][d] * Reg[lane row][d]     (32x32 MFMA tile: the TILE row
//                   lands on the accumulator register index, the wave's own row on the LANE) -> per-lane softmax
//                   statistics need 15 in-register ops + ONE cross-half exchange;
//   tr_accumulate : Y[d, lane row] += sum_{tile row} Tile[row][d] * X[row, lane row]   the fp32 accumulator X is fed
//                   back as the B operand with no data movement (cdna guide section 3, "An accumulator tile as the
//                   next MFMA's operand": element j of lane half h of k-step s is row 16s + 8(j>>2) + 4h + (j&3)),
//                   and Tile^T comes from ds_read_b64_tr_b16 (hardware transpose) on a [row][d] image whose row
//                   stride makes those reads conflict-free (tools/lds_conflicts.py).
//   bf16: v_mfma_f32_32x32x16_bf16, fp32 accumulate.   f32: v_mfma_f32_32x32x2_f32 (exact fp32 fmaf chain).
//
//   forward   (grid q-tile x head x chunk): lane = query; S^T = K.Q^T, online softmax, O^T += V^T.P^T
//   backward A (same grid)                : lane = query; dQ^T += K^T.dS^T          with dS = P o (dP - delta)
//   backward B (grid key-tile x head x chunk): lane = key; dV^T += dO^T.P, dK^T += Q^T.dS   (no atomics: P is
//                   recomputed in both kernels instead of summing dQ across workgroups)
#include "common.h"

namespace {

constexpr int BQ = 128, BKV = 64;
#ifndef ATTN_FWD_OCC
#define ATTN_FWD_OCC 2
#endif

template <typename T, int D> struct ACfg;
template <int D> struct ACfg<bf16_t, D> {
    static constexpr int EPV = 8;                                           // elements per 16-byte chunk
    static constexpr int RSTR = D * 2 + 16;                                 // row-read image stride (bytes)
    static constexpr int TSTR = D == 32 ? 64 : (D == 64 ? 192 : 320);       // transposed-read image stride
    static constexpr bool TWO_IMAGES = true;
};
template <int D> struct ACfg<float, D> {
    static constexpr int EPV = 4;
    static constexpr int RSTR = D * 4 + 16;
    static constexpr int TSTR = D * 4 + 16;                                 // f32: one image serves both reads
    static constexpr bool TWO_IMAGES = false;
};

__device__ __forceinline__ int kappa(int r, int hh) { return (r & 3) + 8 * (r >> 2) + 4 * hh; }

// per-lane operand fragments of one [D] row
template <typename T, int D> struct RowFrag;
template <int D> struct RowFrag<bf16_t, D> {
    bf16x8 v[D / 16];
    __device__ __forceinline__ void load(const bf16_t* row, int hh) {
#pragma unroll
        for (int s = 0; s < D / 16; ++s) v[s] = *(const bf16x8*)(row + 16 * s + 8 * hh);
    }
};
template <int D> struct RowFrag<float, D> {
    f32x4 v[D / 8];
    __device__ __forceinline__ void load(const float* row, int hh) {
#pragma unroll
        for (int u = 0; u < D / 8; ++u) v[u] = *(const f32x4*)(row + 8 * u + 4 * hh);
    }
};

// acc[r] = sum_d Tile[32*sub + kappa(r,hh)][d] * frag(lane row)[d]
template <typename T, int D>
__device__ __forceinline__ f32x16 rows_dot_regs(const unsigned char* img, int sub, const RowFrag<T, D>& f, int lane) {
    using C = ACfg<T, D>;
    const int li = lane & 31, hh = lane >> 5;
    f32x16 s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    const unsigned char* row = img + (sub * 32 + li) * C::RSTR;
    if constexpr (sizeof(T) == 2) {
#pragma unroll
        for (int ss = 0; ss < D / 16; ++ss) {
            bf16x8 a = *(const bf16x8*)(row + (16 * ss + 8 * hh) * 2);
            s = STDIFF_MFMA_32x32x16_H16(a, f.v[ss], s, 0, 0, 0);
        }
    } else {
#pragma unroll
        for (int u = 0; u < D / 8; ++u) {
            f32x4 a = *(const f32x4*)(row + (8 * u + 4 * hh) * 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) s = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], f.v[u][i], s, 0, 0, 0);
        }
    }
    return s;
}

// o[b][r] += sum_{i<32} Tile[32*sub + i][32b + kappa(r,hh)] * x[i][lane row],  x given in accumulator layout
template <typename T, int D>
__device__ __forceinline__ void tr_accumulate(f32x16 (&o)[D / 32], const unsigned char* img, int sub, const f32x16& x,
                                              int lane) {
    using C = ACfg<T, D>;
    const int li = lane & 31, hh = lane >> 5;
    if constexpr (sizeof(T) == 2) {
        bf16x8 xb[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) xb[ks][e] = (h16)x[8 * ks + e];
        const int g1 = (lane >> 4) & 1, q4 = (lane & 15) >> 2, p4 = lane & 3;
#pragma unroll
        for (int b = 0; b < D / 32; ++b) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int kb = sub * 32 + 16 * ks + 4 * hh + q4;
                const unsigned char* va = img + kb * C::TSTR + (b * 32 + 16 * g1 + 4 * p4) * 2;
                bf16x4 lo = ds_read_tr16_b64(va);
                bf16x4 hi = ds_read_tr16_b64(va + 8 * C::TSTR);
                bf16x8 a = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                o[b] = STDIFF_MFMA_32x32x16_H16(a, xb[ks], o[b], 0, 0, 0);
            }
        }
    } else {
#pragma unroll
        for (int b = 0; b < D / 32; ++b) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float a = *(const float*)(img + (sub * 32 + kappa(r, hh)) * C::TSTR + (b * 32 + li) * 4);
                o[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, x[r], o[b], 0, 0, 0);
            }
        }
    }
}

// bf16 only: the two halves of tr_accumulate, so that the V^T fragments of a key tile can be requested BEFORE the
// softmax arithmetic (they do not depend on it) and land under it instead of in front of every MFMA.
template <int D>
__device__ __forceinline__ void tr_fragments(bf16x8 (&a)[D / 32][2], const unsigned char* img, int sub, int lane) {
    using C = ACfg<bf16_t, D>;
    const int hh = lane >> 5, g1 = (lane >> 4) & 1, q4 = (lane & 15) >> 2, p4 = lane & 3;
#pragma unroll
    for (int b = 0; b < D / 32; ++b)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int kb = sub * 32 + 16 * ks + 4 * hh + q4;
            const unsigned char* va = img + kb * C::TSTR + (b * 32 + 16 * g1 + 4 * p4) * 2;
            bf16x4 lo = ds_read_tr16_b64(va);
            bf16x4 hi = ds_read_tr16_b64(va + 8 * C::TSTR);
            a[b][ks] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
}
template <int D>
__device__ __forceinline__ void tr_mma(f32x16 (&o)[D / 32], const bf16x8 (&a)[D / 32][2], const f32x16& x) {
    bf16x8 xb[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) xb[ks][e] = (h16)x[8 * ks + e];
#pragma unroll
    for (int b = 0; b < D / 32; ++b)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) o[b] = STDIFF_MFMA_32x32x16_H16(a[b][ks], xb[ks], o[b], 0, 0, 0);
}

// cooperative staging of a [64][D] tile (rows r0+k0.., clamped to the chunk) into a row image and/or a tr image
template <typename T, int D> struct TileStage {
    using C = ACfg<T, D>;
    static constexpr int CPR = D / C::EPV, LPT = BKV * CPR / 256;
    u32x4 r[LPT];
    __device__ __forceinline__ void gload(const T* base, int ld, int row0, int k0, int len, int tid) {
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            int c = tid + 256 * i, row = c / CPR, ch = c % CPR;
            r[i] = *(const u32x4*)(base + (size_t)(row0 + min(k0 + row, len - 1)) * ld + ch * C::EPV);
        }
    }
    __device__ __forceinline__ void sstore(unsigned char* rimg, unsigned char* timg, int tid) {
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            int c = tid + 256 * i, row = c / CPR, ch = c % CPR;
            if (rimg) *(u32x4*)(rimg + row * C::RSTR + ch * 16) = r[i];
            if (C::TWO_IMAGES && timg) *(u32x4*)(timg + row * C::TSTR + ch * 16) = r[i];
        }
    }
};

template <typename T, int D>
__device__ __forceinline__ void store_rows(T* dst, const f32x16 (&o)[D / 32], float mul, int hh) {
    // lane (row, hh) holds out[row][32b + 8g + 4hh + 0..3]
#pragma unroll
    for (int b = 0; b < D / 32; ++b)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 v = {o[b][4 * g] * mul, o[b][4 * g + 1] * mul, o[b][4 * g + 2] * mul, o[b][4 * g + 3] * mul};
            st4(dst + b * 32 + 8 * g + 4 * hh, v);
        }
}

// ================================================================ forward
template <typename T, int D>
__global__ __launch_bounds__(256, ATTN_FWD_OCC) void attn_fwd_kernel(const T* __restrict__ qkv, int ldqkv, T* __restrict__ out,
                                                       int ldo, float* __restrict__ lse,
                                                       const int32_t* __restrict__ cu_rows, int heads, float c_log2) {
    using C = ACfg<T, D>;
    constexpr bool BF = sizeof(T) == 2;
    // two tile buffers: tile j + 1 is stored while slower waves still read tile j, one barrier per tile
    __shared__ __attribute__((aligned(16))) unsigned char kls2[2][BKV * C::RSTR];
    __shared__ __attribute__((aligned(16))) unsigned char vls2[2][BKV * C::TSTR];

    // XCD-aware order: workgroups go to the 8 XCDs round-robin in linear order, so with the natural (query tile
    // fastest) order the query tiles of one (chunk, head) -- which all read the same K and V rows -- land on
    // different L2s.  Remap: linear id L -> XCD x = L % 8 keeps unit (chunk, head) 8 * (j / gx) + x, query tile j % gx
    // (j = L / 8), so one unit's tiles share an L2 and run back to back.  Needs (heads * chunks) % 8 == 0.
    int chunk = blockIdx.z, hd = blockIdx.y, qt = blockIdx.x;
    {
        const int gx = gridDim.x, units = gridDim.y * gridDim.z;
        if ((units & 7) == 0) {
            const int L = blockIdx.x + gx * (blockIdx.y + gridDim.y * blockIdx.z);
            const int j = L >> 3, unit = ((j / gx) << 3) + (L & 7);
            qt = j % gx;
            hd = unit % gridDim.y;
            chunk = unit / gridDim.y;
        }
    }
    const int r0 = cu_rows[chunk], len = cu_rows[chunk + 1] - r0;
    const int q0 = qt * BQ;
    if (q0 >= len) return;
    const int H = heads * D;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int qi = lane & 31, hh = lane >> 5;
    const int qrow = q0 + wid * 32 + qi;

    RowFrag<T, D> qf;
    qf.load(qkv + (size_t)(r0 + min(qrow, len - 1)) * ldqkv + hd * D, hh);

    f32x16 o[D / 32];
#pragma unroll
    for (int b = 0; b < D / 32; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[b][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    TileStage<T, D> sk, sv;
    const T* kbase = qkv + H + hd * D;
    const T* vbase = qkv + 2 * H + hd * D;
    const int ntiles = (len + BKV - 1) / BKV;
    if constexpr (BF) { sk.gload(kbase, ldqkv, r0, 0, len, tid); sv.gload(vbase, ldqkv, r0, 0, len, tid); }
    for (int j = 0; j < ntiles; ++j) {
        const int k0 = j * BKV;
        unsigned char* kls = kls2[j & 1];
        unsigned char* vls = vls2[j & 1];
        if constexpr (!BF) { sk.gload(kbase, ldqkv, r0, k0, len, tid); sv.gload(vbase, ldqkv, r0, k0, len, tid); }
        sk.sstore(kls, nullptr, tid);
        if constexpr (BF) sv.sstore(nullptr, vls, tid); else sv.sstore(vls, nullptr, tid);
        __syncthreads();
        if constexpr (BF) {
            if (j + 1 < ntiles) { sk.gload(kbase, ldqkv, r0, k0 + BKV, len, tid); sv.gload(vbase, ldqkv, r0, k0 + BKV, len, tid); }
        }
        // 64 keys per softmax step (two 32-key MFMA tiles).  The running maximum is raised LAZILY: only when some
        // lane's new maximum exceeds its reference by more than 2^LAZY does the wave rescale O and l (a skipped
        // rescale multiplies by exactly 1, so nothing changes but the work: the accumulators stay in place instead of
        // a read-modify-write of all of them per tile).  exp2 is the bare v_exp_f32.
        constexpr float LAZY = 6.0f;
        constexpr bool PREV = BF && D <= 64;                         // registers allow it (d = 128: 226 VGPRs already)
        bf16x8 vf[2][PREV ? D / 32 : 1][2];
        if constexpr (PREV) {
            tr_fragments<D>(vf[0], vls, 0, lane);
            tr_fragments<D>(vf[1], vls, 1, lane);                    // a tile past the chunk: finite stale data x P = 0
            __builtin_amdgcn_sched_barrier(0);
        }
        // S^T[key, q], raw scores.  Both 32-key halves always: the staged rows past the chunk are copies of its last
        // row (finite), the mask below turns their scores into -inf, so P = 0 there and the V MFMAs add exact zeros
        f32x16 sa = rows_dot_regs<T, D>(kls, 0, qf, lane);
        f32x16 sb = rows_dot_regs<T, D>(kls, 1, qf, lane);
        if (k0 + 64 > len) {                                         // keys past the chunk get -inf
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (k0 + kappa(r, hh) >= len) sa[r] = -INFINITY;
                if (k0 + 32 + kappa(r, hh) >= len) sb[r] = -INFINITY;
            }
        }
        float mt = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) mt = fmaxf(mt, fmaxf(sa[r], sb[r]));
        mt = fmaxf(mt, __shfl_xor(mt, 32, 64)) * c_log2;
        if (__builtin_amdgcn_ballot_w64(mt > m_run + LAZY) != 0) {
            const float m_new = fmaxf(m_run, mt);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            l_run *= alpha;
            m_run = m_new;
#pragma unroll
            for (int b = 0; b < D / 32; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[b][r] *= alpha;
        }
        float ps = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            sa[r] = __builtin_amdgcn_exp2f(fmaf(sa[r], c_log2, -m_run));
            sb[r] = __builtin_amdgcn_exp2f(fmaf(sb[r], c_log2, -m_run));
            ps += sa[r] + sb[r];
        }
        l_run += ps;
        if constexpr (PREV) {
            tr_mma<D>(o, vf[0], sa);                                 // O^T[d, q] += V^T . P^T
            tr_mma<D>(o, vf[1], sb);
        } else {
            tr_accumulate<T, D>(o, vls, 0, sa, lane);
            tr_accumulate<T, D>(o, vls, 1, sb, lane);
        }
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    if (qrow < len) {
        store_rows<T, D>(out + (size_t)(r0 + qrow) * ldo + hd * D, o, 1.0f / l_tot, hh);
        if (lse && hh == 0) lse[(size_t)(r0 + qrow) * heads + hd] = m_run + log2f(l_tot);
    }
}

// ================================================================ backward
// delta[r, h] = sum_d dO[r, hD+d] * O[r, hD+d]
// SPLIT: dout still lies as split-K partial products of the projection's input-gradient GEMM (stdiff_split_result): the
// wave sums its piece in slice order, rounds it to T, WRITES it to dout (the dQ / dK / dV passes read it from there) and
// goes on with the rounded values -- the bits the reduce launch it replaces would have left.
template <typename T, bool SPLIT = false>
__global__ __launch_bounds__(256) void attn_delta_kernel(const T* o, int ldo, T* dout, int lddo, float* delta,
                                                         int R, int heads, int D, const float* ws, int ldws, int slices,
                                                         long long stride) {
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6);      // one wave per (row, head)
    if (idx >= R * heads) return;
    const int r = idx / heads, h = idx % heads, lane = threadIdx.x & 63;
    float s = 0.f;
    for (int c = lane * 4; c < D; c += 256) {
        f32x4 a = ld4(o + (size_t)r * ldo + h * D + c), b;
        if constexpr (SPLIT) {
            const float* w = ws + (size_t)r * ldws + h * D + c;
            b = sum_slices(w, slices, stride);
            st4(dout + (size_t)r * lddo + h * D + c, b);
            if constexpr (sizeof(T) == 2) {
#pragma unroll
                for (int j = 0; j < 4; ++j) b[j] = bf2f(f2bf(b[j]));
            }
        } else {
            b = ld4(dout + (size_t)r * lddo + h * D + c);
        }
        s += (a[0] * b[0] + a[1] * b[1]) + (a[2] * b[2] + a[3] * b[3]);
    }
    s = wave_sum(s);
    if (lane == 0) delta[idx] = s;
}

// A: dQ.  lane = query row; loops over key tiles.
template <typename T, int D>
__device__ __forceinline__ void attn_bwd_dq_body(const int bx, const T* __restrict__ qkv, int ldqkv,
                                                 const T* __restrict__ dout, int lddo,
                                                 const float* __restrict__ lse,
                                                 const float* __restrict__ delta, T* __restrict__ dqkv,
                                                 int lddqkv, const int32_t* __restrict__ cu_rows, int heads,
                                                 float c_log2, float scale) {
    using C = ACfg<T, D>;
    __shared__ __attribute__((aligned(16))) unsigned char k_r[BKV * C::RSTR];
    __shared__ __attribute__((aligned(16))) unsigned char k_t[C::TWO_IMAGES ? BKV * C::TSTR : 16];
    __shared__ __attribute__((aligned(16))) unsigned char v_r[BKV * C::RSTR];

    const int chunk = blockIdx.z, hd = blockIdx.y;
    const int r0 = cu_rows[chunk], len = cu_rows[chunk + 1] - r0;
    const int q0 = bx * BQ;
    if (q0 >= len) return;
    const int H = heads * D;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int qi = lane & 31, hh = lane >> 5;
    const int qrow = q0 + wid * 32 + qi, qc = min(qrow, len - 1);

    RowFrag<T, D> qf, dof;
    qf.load(qkv + (size_t)(r0 + qc) * ldqkv + hd * D, hh);
    dof.load(dout + (size_t)(r0 + qc) * lddo + hd * D, hh);
    const float my_lse = lse[(size_t)(r0 + qc) * heads + hd], my_delta = delta[(size_t)(r0 + qc) * heads + hd];

    f32x16 dq[D / 32];
#pragma unroll
    for (int b = 0; b < D / 32; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[b][r] = 0.f;

    TileStage<T, D> sk, sv;
    const T* kbase = qkv + H + hd * D;
    const T* vbase = qkv + 2 * H + hd * D;
    const unsigned char* ktr = C::TWO_IMAGES ? k_t : k_r;
    const int ntiles = (len + BKV - 1) / BKV;
    for (int j = 0; j < ntiles; ++j) {
        const int k0 = j * BKV;
        __syncthreads();
        sk.gload(kbase, ldqkv, r0, k0, len, tid);
        sv.gload(vbase, ldqkv, r0, k0, len, tid);
        sk.sstore(k_r, k_t, tid);
        sv.sstore(v_r, nullptr, tid);
        __syncthreads();
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            if (k0 + kt * 32 >= len) break;
            f32x16 s = rows_dot_regs<T, D>(k_r, kt, qf, lane);       // S^T[key, q]
            f32x16 dp = rows_dot_regs<T, D>(v_r, kt, dof, lane);     // dP^T[key, q] = V . dO^T
            const bool tail = k0 + kt * 32 + 32 > len;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float p = exp2f(s[r] * c_log2 - my_lse);
                if (tail && k0 + kt * 32 + kappa(r, hh) >= len) p = 0.f;
                s[r] = p * (dp[r] - my_delta);                       // dS^T
            }
            tr_accumulate<T, D>(dq, ktr, kt, s, lane);               // dQ^T[d, q] += K^T . dS^T
        }
    }
    if (qrow < len) store_rows<T, D>(dqkv + (size_t)(r0 + qrow) * lddqkv + hd * D, dq, scale, hh);
}

// B: dK, dV.  lane = key row; loops over query tiles.
template <typename T, int D>
__device__ __forceinline__ void attn_bwd_dkv_body(const int bx, const T* __restrict__ qkv, int ldqkv,
                                                  const T* __restrict__ dout, int lddo,
                                                  const float* __restrict__ lse,
                                                  const float* __restrict__ delta, T* __restrict__ dqkv,
                                                  int lddqkv, const int32_t* __restrict__ cu_rows, int heads,
                                                  float c_log2, float scale) {
    using C = ACfg<T, D>;
    __shared__ __attribute__((aligned(16))) unsigned char q_r[BKV * C::RSTR];
    __shared__ __attribute__((aligned(16))) unsigned char q_t[C::TWO_IMAGES ? BKV * C::TSTR : 16];
    __shared__ __attribute__((aligned(16))) unsigned char d_r[BKV * C::RSTR];
    __shared__ __attribute__((aligned(16))) unsigned char d_t[C::TWO_IMAGES ? BKV * C::TSTR : 16];
    __shared__ float lse_s[BKV], delta_s[BKV];

    const int chunk = blockIdx.z, hd = blockIdx.y;
    const int r0 = cu_rows[chunk], len = cu_rows[chunk + 1] - r0;
    const int kbase0 = bx * BQ;
    if (kbase0 >= len) return;
    const int H = heads * D;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int ki = lane & 31, hh = lane >> 5;
    const int krow = kbase0 + wid * 32 + ki, kc = min(krow, len - 1);

    RowFrag<T, D> kf, vf;
    kf.load(qkv + (size_t)(r0 + kc) * ldqkv + H + hd * D, hh);
    vf.load(qkv + (size_t)(r0 + kc) * ldqkv + 2 * H + hd * D, hh);

    f32x16 dk[D / 32], dv[D / 32];
#pragma unroll
    for (int b = 0; b < D / 32; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) { dk[b][r] = 0.f; dv[b][r] = 0.f; }

    TileStage<T, D> sq, sd;
    const T* qbase = qkv + hd * D;
    const T* dbase = dout + hd * D;
    const unsigned char* qtr = C::TWO_IMAGES ? q_t : q_r;
    const unsigned char* dtr = C::TWO_IMAGES ? d_t : d_r;
    const int ntiles = (len + BKV - 1) / BKV;
    for (int j = 0; j < ntiles; ++j) {
        const int t0 = j * BKV;
        __syncthreads();
        sq.gload(qbase, ldqkv, r0, t0, len, tid);
        sd.gload(dbase, lddo, r0, t0, len, tid);
        sq.sstore(q_r, q_t, tid);
        sd.sstore(d_r, d_t, tid);
        if (tid < BKV) {
            const int q = t0 + tid;
            lse_s[tid] = q < len ? lse[(size_t)(r0 + q) * heads + hd] : INFINITY;     // +inf -> P = 0 past the chunk
            delta_s[tid] = q < len ? delta[(size_t)(r0 + q) * heads + hd] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
            if (t0 + qt * 32 >= len) break;
            f32x16 s = rows_dot_regs<T, D>(q_r, qt, kf, lane);       // S[q, key]
            f32x16 dp = rows_dot_regs<T, D>(d_r, qt, vf, lane);      // dP[q, key] = dO . V^T
            f32x16 ds;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int q = qt * 32 + kappa(r, hh);
                float p = exp2f(s[r] * c_log2 - lse_s[q]);
                s[r] = p;
                ds[r] = p * (dp[r] - delta_s[q]);
            }
            tr_accumulate<T, D>(dv, dtr, qt, s, lane);               // dV^T[d, key] += dO^T . P
            tr_accumulate<T, D>(dk, qtr, qt, ds, lane);              // dK^T[d, key] += Q^T . dS
        }
    }
    if (krow < len) {
        T* dst = dqkv + (size_t)(r0 + krow) * lddqkv + hd * D;
        store_rows<T, D>(dst + H, dk, scale, hh);
        store_rows<T, D>(dst + 2 * H, dv, 1.0f, hh);
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

// dQ and dK/dV as ONE launch: blocks [0, gx) of grid.x run the dQ pass, blocks [gx, 2 gx) the dK/dV pass (both read q, k,
// v, dO, lse, delta; they write disjoint column ranges of dqkv).  At a 512-cell training batch either pass alone is 64
// workgroups on a 256-CU chip and 23-25 us of latency; side by side they cost one of the two.
template <typename T, int D>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const T* __restrict__ qkv, int ldqkv, const T* __restrict__ dout,
                                                       int lddo, const float* __restrict__ lse,
                                                       const float* __restrict__ delta, T* __restrict__ dqkv, int lddqkv,
                                                       const int32_t* __restrict__ cu_rows, int heads, float c_log2,
                                                       float scale) {
    const int gx = gridDim.x >> 1;
    if ((int)blockIdx.x < gx)
        attn_bwd_dq_body<T, D>(blockIdx.x, qkv, ldqkv, dout, lddo, lse, delta, dqkv, lddqkv, cu_rows, heads, c_log2, scale);
    else
        attn_bwd_dkv_body<T, D>(blockIdx.x - gx, qkv, ldqkv, dout, lddo, lse, delta, dqkv, lddqkv, cu_rows, heads, c_log2, scale);
}

template <typename T>
int launch_bwd(const void* qkv, int ldqkv, const void* out, int ldo, const void* dout_, int lddo, const float* lse,
               float* delta, void* dqkv, int lddqkv, const int32_t* cu, int n_chunks, int maxc, int R, int heads, int d,
               hipStream_t st, const stdiff_split_result* sp = nullptr) {
    void* dout = const_cast<void*>(dout_);          // written only when it is formed here from `sp`
    if (sp)
        hipLaunchKernelGGL((attn_delta_kernel<T, true>), dim3(cdiv((long)R * heads, 4)), dim3(256), 0, st, (const T*)out,
                           ldo, (T*)dout, lddo, delta, R, heads, d, sp->ws, sp->ldws, sp->slices, sp->slice_stride);
    else
        hipLaunchKernelGGL((attn_delta_kernel<T, false>), dim3(cdiv((long)R * heads, 4)), dim3(256), 0, st, (const T*)out,
                           ldo, (T*)dout, lddo, delta, R, heads, d, (const float*)nullptr, 0, 0, 0ll);
    dim3 grid(2 * cdiv(maxc, BQ), heads, n_chunks), block(256);
    const float scale = 1.0f / sqrtf((float)d), c = scale * 1.4426950408889634f;
#define CASE(DD)                                                                                                  \
    hipLaunchKernelGGL((attn_bwd_kernel<T, DD>), grid, block, 0, st, (const T*)qkv, ldqkv, (const T*)dout, lddo, lse,  \
                       delta, (T*)dqkv, lddqkv, cu, heads, c, scale)
    if (d == 32) CASE(32);
    else if (d == 64) CASE(64);
    else if (d == 128) CASE(128);
    else return STDIFF_ERR_SHAPE;
#undef CASE
    STDIFF_LAUNCH_STATUS();
    return STDIFF_OK;
}

int check_common(const void* qkv, int ldqkv, int ldo, const int32_t* cu_rows, int n_chunks, int maxc, int heads,
                 int head_dim, int dtype) {
    STDIFF_CHECK_ARG(qkv && cu_rows && n_chunks >= 0 && maxc >= 0 && heads > 0);
    STDIFF_CHECK_ARG(dtype == STDIFF_F32 || dtype == STDIFF_BF16);
    if (head_dim != 32 && head_dim != 64 && head_dim != 128) return STDIFF_ERR_SHAPE;
    if (heads > 65535 || n_chunks > 65535) return STDIFF_ERR_SHAPE;
    const int epv = dtype == STDIFF_BF16 ? 8 : 4;
    if (ldqkv % epv || ldo % epv || ((uintptr_t)qkv & 15)) return STDIFF_ERR_ALIGN;
    if (ldqkv < 3 * heads * head_dim || ldo < heads * head_dim) return STDIFF_ERR_ARG;
    return STDIFF_OK;
}

}  // namespace

extern "C" int stdiff_attention_fwd(const void* qkv, int ldqkv, void* out, int ldo, float* lse, const int32_t* cu_rows,
                                    int n_chunks, int max_chunk_rows, int heads, int head_dim, int dtype,
                                    void* stream) {
    STDIFF_CHECK_ARG(out != nullptr);
    int rc = check_common(qkv, ldqkv, ldo, cu_rows, n_chunks, max_chunk_rows, heads, head_dim, dtype);
    if (rc != STDIFF_OK) return rc;
    if ((uintptr_t)out & 15) return STDIFF_ERR_ALIGN;
    if (n_chunks == 0 || max_chunk_rows == 0) return STDIFF_OK;
    if (dtype == STDIFF_BF16)
        return launch_fwd<bf16_t>(qkv, ldqkv, out, ldo, lse, cu_rows, n_chunks, max_chunk_rows, heads, head_dim,
                                  (hipStream_t)stream);
    return launch_fwd<float>(qkv, ldqkv, out, ldo, lse, cu_rows, n_chunks, max_chunk_rows, heads, head_dim,
                             (hipStream_t)stream);
}

extern "C" int stdiff_attention_bwd(const void* qkv, int ldqkv, const void* out, int ldo, const void* dout, int lddo,
                                    const float* lse, float* delta, void* dqkv, int lddqkv, const int32_t* cu_rows,
                                    int n_chunks, int max_chunk_rows, int total_rows, int heads, int head_dim,
                                    int dtype, void* stream) {
    STDIFF_CHECK_ARG(out && dout && lse && delta && dqkv && total_rows >= 0);
    int rc = check_common(qkv, ldqkv, ldo, cu_rows, n_chunks, max_chunk_rows, heads, head_dim, dtype);
    if (rc != STDIFF_OK) return rc;
    const int epv = dtype == STDIFF_BF16 ? 8 : 4;
    if (lddo % epv || lddqkv % epv || ((uintptr_t)out & 15) || ((uintptr_t)dout & 15) || ((uintptr_t)dqkv & 15))
        return STDIFF_ERR_ALIGN;
    if (lddo < heads * head_dim || lddqkv < 3 * heads * head_dim) return STDIFF_ERR_ARG;
    if (n_chunks == 0 || max_chunk_rows == 0 || total_rows == 0) return STDIFF_OK;
    if (dtype == STDIFF_BF16)
        return launch_bwd<bf16_t>(qkv, ldqkv, out, ldo, dout, lddo, lse, delta, dqkv, lddqkv, cu_rows, n_chunks,
                                  max_chunk_rows, total_rows, heads, head_dim, (hipStream_t)stream);
    return launch_bwd<float>(qkv, ldqkv, out, ldo, dout, lddo, lse, delta, dqkv, lddqkv, cu_rows, n_chunks,
                             max_chunk_rows, total_rows, heads, head_dim, (hipStream_t)stream);
}

extern "C" int stdiff_attention_bwd_split(const void* qkv, int ldqkv, const void* out, int ldo,
                                          const stdiff_split_result* sp, void* dout, int lddo, const float* lse,
                                          float* delta, void* dqkv, int lddqkv, const int32_t* cu_rows, int n_chunks,
                                          int max_chunk_rows, int total_rows, int heads, int head_dim, int dtype,
                                          void* stream) {
    STDIFF_CHECK_ARG(sp && sp->ws && sp->slices >= 1 && sp->slices <= STDIFF_MAX_SLICES && sp->ldws >= heads * head_dim);
    STDIFF_CHECK_ARG(out && dout && lse && delta && dqkv && total_rows >= 0);
    int rc = check_common(qkv, ldqkv, ldo, cu_rows, n_chunks, max_chunk_rows, heads, head_dim, dtype);
    if (rc != STDIFF_OK) return rc;
    const int epv = dtype == STDIFF_BF16 ? 8 : 4;
    if (lddo % epv || lddqkv % epv || ((uintptr_t)out & 15) || ((uintptr_t)dout & 15) || ((uintptr_t)dqkv & 15) ||
        sp->ldws % 4 || sp->slice_stride % 4 || ((uintptr_t)sp->ws & 15))
        return STDIFF_ERR_ALIGN;
    if (lddo < heads * head_dim || lddqkv < 3 * heads * head_dim) return STDIFF_ERR_ARG;
    if (n_chunks == 0 || max_chunk_rows == 0 || total_rows == 0) return STDIFF_OK;
    if (dtype == STDIFF_BF16)
        return launch_bwd<bf16_t>(qkv, ldqkv, out, ldo, dout, lddo, lse, delta, dqkv, lddqkv, cu_rows, n_chunks,
                                  max_chunk_rows, total_rows, heads, head_dim, (hipStream_t)stream, sp);
    return launch_bwd<float>(qkv, ldqkv, out, ldo, dout, lddo, lse, delta, dqkv, lddqkv, cu_rows, n_chunks,
                             max_chunk_rows, total_rows, heads, head_dim, (hipStream_t)stream, sp);
}
