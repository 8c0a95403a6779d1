// NT GEMM on the gfx950 matrix cores:  C[M,N] = epi(A[M,K] . W[N,K]^T + bias[N]).
//
// Every dense contraction of the denoiser (nn.Linear: stDiff_model.py:18,20,54,56,107,128,132,177,180 and the
// timm Mlp fc1/fc2) is this shape in the forward pass: both operands K-contiguous, both tiles staged as
// [rows][K-slab], both MFMA fragments plain 16-byte LDS reads.  MFMA is issued "swapped" (the W fragment as the A
// operand) so a lane ends up holding 4 CONSECUTIVE n of one m.  The backward contractions (dW = dY^T X, dX = dY W)
// are the same product with one or both operands K-MAJOR; the 128 x 128 kernel reads those as they lie.
//
// Kernels:
//   gemm256q_kernel  16-bit operands (bf16, or fp16 in the -DSTDIFF_H16_FP16 build), 256 x 256 block tile, 8 waves,
//                    persistent (one workgroup per CU walks the tiles), LDS-DMA staging, 2 phases of 32 MFMAs per
//                    K-slab with counted waits and two staggered wave groups.  The default when the output has
//                    >= 128 tiles.  gemm256p_kernel is its 4-phase predecessor (STDIFF_GEMM_TILE=3: A/B timing and
//                    a second implementation for the tests).
//   gemm_nt_kernel   bf16 or exact-fp32, 128 x 128 tile, 4 waves, register-staged double buffer, 2 blocks per CU;
//                    optional K-major operands (transposed LDS reads) and deterministic split-K through a caller
//                    workspace.  bf16: v_mfma_f32_16x16x32_bf16;  f32: v_mfma_f32_16x16x4_f32 (a k-ordered fp32
//                    fmaf chain).
//   gemm128d_kernel  the same 128 x 128 tile for 16-bit row-major operands on grids of at most one workgroup per CU (the
//                    batch-sized forward GEMMs of a training step): LDS-DMA staging into a ring of four slab buffers,
//                    eight waves, one barrier per slab; the same bits as gemm_nt_kernel.
// All of them send the accumulators through an fp32 LDS image of the C tile so that the epilogue (bias / SiLU / GELU-erf /
// gate*v+res / +res / accumulate / save-pre-activation) loads and stores whole row pieces, in batches whose latencies
// overlap.  The epilogue math is a compile-time template parameter dispatched ONCE per block: a runtime switch inside
// the unrolled store loop replicated the erf/exp code 32x (128 KiB of instructions) and cost 13 us per tile.
// Grid: 1-D, XCD-aware (blocks b and b+8 share an L2): each XCD walks a contiguous range of tiles, ordered in groups
// of 8 row-tiles x all column-tiles so the A panel and the weight panel stay L2-resident.
#include <math.h>
#include <stdlib.h>

#include <type_traits>

#include "common.h"

namespace {

constexpr int BM = 128, BN = 128, NT = 256;
constexpr int CLD = BN + 4;  // fp32 C image row stride (floats): +16 B keeps ds_write_b128 conflict-free

template <typename T> struct Cfg;
template <> struct Cfg<bf16_t> {
    static constexpr int BK = 64, EPV = 8, NSLOT = 8, ROWB = 128;
    static __device__ __forceinline__ int swz(int row) { return row & 7; }
};
template <> struct Cfg<float> {
    static constexpr int BK = 16, EPV = 4, NSLOT = 4, ROWB = 64;
    static __device__ __forceinline__ int swz(int row) { return (row >> 1) & 3; }
};

struct Params {
    int M, N, K;
    const void* A; int lda;
    const void* W; int ldw;
    const float* bias;
    void* C; int ldc; int c_bf16;
    int epi;
    const void* gate; int ldgate;
    const float* res; int ldres;
    void* C2; int ldc2;
    int vec;  // != 0: every epilogue pointer / stride allows aligned 4-wide (16 B fp32, 8 B bf16) access and N % 8 == 0;
              // 2: a bf16 C additionally allows 16-byte row pieces
    int tiles_m, tiles_n;
    int k_valid;         // K-major operands: rows k >= k_valid are zero (K itself is padded to the slab size)
    int ksplit, nk_per;  // split-K (128 x 128 kernel): grid = tiles * ksplit, split s covers K-slabs [s*nk_per, ..)
    float* ws; int ldws; // partial products [ksplit][M][ldws] fp32
    // gemm256q work units: units [0, t_full) are whole 256 x 256 tiles; unit t_full + 2j + h is the 128-row half h of
    // tile t_full + j (the tiles of a short last round are cut in two so that the round costs about half)
    int t_full, units;
    const float* ln_parts; int ln_nparts;         // STDIFF_EPI_LN_MOD: partial row moments of `res` [M][parts][2]
    float* stats_out;                             // gate-residual epilogue: partial row moments of C [M][tiles_n][2]
    float* colsum;                                // gemm_dw_group_kernel only: sum over k of A[m][k]  (fp32 [M])
    float* sumsq;                                 // gemm_dw_group_kernel only: the sum of squares of what the tile stores
};

__device__ __forceinline__ void tile_of(int bid, int nblk, int tiles_m, int tiles_n, int& tm, int& tn) {
    // bijective XCD remap (cdna guide T1): blocks with equal bid%8 share an XCD -> give each a contiguous range
    int q = nblk >> 3, r = nblk & 7, x = bid & 7;
    int pid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
    constexpr int GM = 8;
    int per = GM * tiles_n;
    int grp = pid / per, in = pid - grp * per;
    int first = grp * GM;
    int gsz = min(tiles_m - first, GM);
    tm = first + in % gsz;
    tn = in / gsz;
}

// ---------------------------------------------------------------- epilogue pieces (4 consecutive columns of one row)
template <int EPI, bool FAST = false> __device__ __forceinline__ f32x4 epi_math(f32x4 pre, f32x4 g, f32x4 r) {
    f32x4 o;
    if constexpr (FAST && (EPI == STDIFF_EPI_GELU || EPI == STDIFF_EPI_GELU_SAVE)) {
        o = gelu_poly4(pre);            // bf16 result: the polynomial's 1.3e-6 is far below the output rounding
    } else if constexpr (EPI == STDIFF_EPI_SILU) {
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = silu_f(pre[j]);
    } else if constexpr (EPI == STDIFF_EPI_GELU || EPI == STDIFF_EPI_GELU_SAVE) {
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = gelu_erf_f(pre[j]);
    } else if constexpr (EPI == STDIFF_EPI_GATE_RES) {
        o = r + g * pre;
    } else if constexpr (EPI == STDIFF_EPI_ADD || EPI == STDIFF_EPI_ACCUM) {
        o = pre + r;
    } else if constexpr (FAST && EPI == STDIFF_EPI_GELU_BWD) {      // 16-bit result: no erff / expf calls
        o = pre * gelu_grad4(g);
    } else if constexpr (EPI == STDIFF_EPI_GELU_BWD) {      // g = the saved pre-activation; stdiff_gelu_bwd's expression
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float x = g[j];
            const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
            const float pdf = 0.3989422804014327f * expf(-0.5f * x * x);
            o[j] = pre[j] * (cdf + x * pdf);
        }
    } else {
        o = pre;
    }
    return o;
}

// aligned side loads for one piece (row / column already clamped into range by the caller)
template <int EPI, typename T>
__device__ __forceinline__ void epi_load(const Params& p, int gm, int gn, f32x4& g, f32x4& r) {
    if constexpr (EPI == STDIFF_EPI_GATE_RES || EPI == STDIFF_EPI_GELU_BWD) g = ld4((const T*)p.gate + (size_t)gm * p.ldgate + gn);
    if constexpr (EPI == STDIFF_EPI_GATE_RES || EPI == STDIFF_EPI_ADD) r = *(const f32x4*)(p.res + (size_t)gm * p.ldres + gn);
    if constexpr (EPI == STDIFF_EPI_ACCUM) r = *(const f32x4*)((const float*)p.C + (size_t)gm * p.ldc + gn);
}

template <int EPI, typename T>
__device__ __forceinline__ void epi_store(const Params& p, int gm, int gn, f32x4 pre, f32x4 g, f32x4 r) {
    // bf16 outputs take the polynomial GELU here too: every bf16 kernel then produces the same bits for the same
    // row, whichever tile shape the host heuristic picks (chunk-, lane- and shard-invariance of the sampler)
    if constexpr (EPI == STDIFF_EPI_GELU_BWD) {      // the unfused pair rounds the GEMM result to the output dtype first
        if (p.c_bf16) {
#pragma unroll
            for (int j = 0; j < 4; ++j) pre[j] = bf2f(f2bf(pre[j]));
        }
    }
    if (p.c_bf16) st4((bf16_t*)p.C + (size_t)gm * p.ldc + gn, epi_math<EPI, true>(pre, g, r));
    else *(f32x4*)((float*)p.C + (size_t)gm * p.ldc + gn) = epi_math<EPI, false>(pre, g, r);
    if (p.C2) st4((T*)p.C2 + (size_t)gm * p.ldc2 + gn, pre);       // v itself, in the operand dtype
}

// any alignment, ragged N: scalar, runtime epilogue; callers keep it inside `#pragma unroll 1` loops so the code
// exists once (odd shapes only, e.g. G = 33 genes)
template <typename T>
__device__ __forceinline__ void epilogue4_slow(const Params& p, int gm, int gn, f32x4 v) {
#pragma unroll 1
    for (int j = 0; j < 4; ++j) {
        const int n = gn + j;
        if (n >= p.N) return;
        const float pre = v[j] + (p.bias ? p.bias[n] : 0.f);
        float o;
        switch (p.epi) {
            case STDIFF_EPI_SILU: o = silu_f(pre); break;
            case STDIFF_EPI_GELU:
            case STDIFF_EPI_GELU_SAVE: o = gelu_erf_f(pre); break;
            case STDIFF_EPI_GATE_RES:
                o = p.res[(size_t)gm * p.ldres + n] + Elem<T>::ld((const T*)p.gate + (size_t)gm * p.ldgate + n) * pre;
                break;
            case STDIFF_EPI_ADD: o = pre + p.res[(size_t)gm * p.ldres + n]; break;
            case STDIFF_EPI_ACCUM: o = pre + ((const float*)p.C)[(size_t)gm * p.ldc + n]; break;
            case STDIFF_EPI_GELU_BWD: {
                const float x = Elem<T>::ld((const T*)p.gate + (size_t)gm * p.ldgate + n);
                const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
                const float pdf = 0.3989422804014327f * expf(-0.5f * x * x);
                o = (p.c_bf16 ? bf2f(f2bf(pre)) : pre) * (cdf + x * pdf);
                break;
            }
            default: o = pre;
        }
        if (p.c_bf16) ((bf16_t*)p.C)[(size_t)gm * p.ldc + n] = f2bf(o);
        else ((float*)p.C)[(size_t)gm * p.ldc + n] = o;
        if (p.C2) Elem<T>::st((T*)p.C2 + (size_t)gm * p.ldc2 + n, pre);
    }
}

#define STDIFF_EPI_DISPATCH(CALL)                                  \
    switch (p.epi) {                                               \
        case STDIFF_EPI_SILU: CALL(STDIFF_EPI_SILU); break;        \
        case STDIFF_EPI_GELU:                                      \
        case STDIFF_EPI_GELU_SAVE: CALL(STDIFF_EPI_GELU); break;   \
        case STDIFF_EPI_GATE_RES: CALL(STDIFF_EPI_GATE_RES); break; \
        case STDIFF_EPI_ADD: CALL(STDIFF_EPI_ADD); break;          \
        case STDIFF_EPI_ACCUM: CALL(STDIFF_EPI_ACCUM); break;      \
        case STDIFF_EPI_GELU_BWD: CALL(STDIFF_EPI_GELU_BWD); break; \
        default: CALL(STDIFF_EPI_BIAS); break;                     \
    }

// ================================================================ 128 x 128 kernel (bf16 / fp32)
// rows of the fp32 C image `cl` [128][CLD] -> global: thread = 8 columns (two pieces) of rows (tid>>4) + 16*pass
template <int EPI, typename T>
__device__ __forceinline__ void epi_rows128(const Params& p, const float* cl, int tid, int m0, int n0) {
    const int c8 = (tid & 15) * 8, r0 = tid >> 4;
    const int gn = n0 + c8;
    if (gn >= p.N) return;
    f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
    if (p.bias) { b0 = *(const f32x4*)(p.bias + gn); b1 = *(const f32x4*)(p.bias + gn + 4); }
#pragma unroll 1
    for (int pb = 0; pb < 2; ++pb) {
        f32x4 g[4][2], r[4][2];
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int gm = min(m0 + r0 + 16 * (pb * 4 + ps), p.M - 1);
            epi_load<EPI, T>(p, gm, gn, g[ps][0], r[ps][0]);
            epi_load<EPI, T>(p, gm, gn + 4, g[ps][1], r[ps][1]);
        }
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int row = r0 + 16 * (pb * 4 + ps), gm = m0 + row;
            if (gm < p.M) {
                const f32x4 v0 = *(const f32x4*)(cl + row * CLD + c8) + b0;
                const f32x4 v1 = *(const f32x4*)(cl + row * CLD + c8 + 4) + b1;
                epi_store<EPI, T>(p, gm, gn, v0, g[ps][0], r[ps][0]);
                epi_store<EPI, T>(p, gm, gn + 4, v1, g[ps][1], r[ps][1]);
            }
        }
    }
}

// AK / WK: the operand is given K-MAJOR ([K][M] resp. [K][N], the non-contracted index contiguous) instead of
// [M][K] / [N][K].  That is how the backward contractions see their operands (dW = dY^T X: both; dX = dY W: W), so
// no transposed copies are made: the tile is staged as [k][m] rows (32-byte / 64-byte column units XOR-swizzled by
// the row) and the MFMA fragments come from ds_read_b64_tr_b16 (bf16: the hardware transposes a 4-row x 16-column
// block per 16 lanes) or scalar LDS reads (fp32).  Rows k >= k_valid are staged as zeros (a batch that is not a
// multiple of the K-slab).
// DEPTH: K-slabs of global loads kept in flight in registers (1 = the slab after the one being multiplied).  Grids of
// at most one block per CU (the training step's batch-sized GEMMs) have nothing else on the CU to hide the load
// latency behind, so they run with DEPTH = 3: a slab's loads are issued three multiplies before its LDS store.
template <typename T> constexpr int smem128() {
    constexpr int stage = 4 * BM * Cfg<T>::ROWB, cbytes = BM * CLD * 4;      // 2 buffers x (A + W)  |  the fp32 C image
    return stage > cbytes ? stage : cbytes;
}

// GROUPED: the body runs as one problem of gemm_dw_group_kernel -- plain fp32 store only, and the tiles of the first
// tile column also sum their A tile over k (p.colsum: the bias gradient of a weight-gradient product, on the matrix
// cores: one more MFMA per row block against a fragment of ones).
template <typename T, bool AK, bool WK, int DEPTH, bool GROUPED>
__device__ __forceinline__ void gemm_nt_body(const Params& p, const int bid, unsigned char* smem) {
    using C = Cfg<T>;
    constexpr int BK = C::BK, EPV = C::EPV, NSLOT = C::NSLOT, ROWB = C::ROWB;
    constexpr int TILEB = BM * ROWB;               // bytes of one operand tile
    constexpr int LPT = BM * NSLOT / NT;           // 16-byte loads per thread per operand

    int tm, tn;
    const int ntiles = p.tiles_m * p.tiles_n;
    const int split = bid / ntiles;
    tile_of(bid - split * ntiles, ntiles, p.tiles_m, p.tiles_n, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;

    // ---- staging addresses (rows past the edge are clamped: their products are never stored)
    constexpr int PPR = BM / EPV;                  // K-major: 16-byte pieces per k-row of the tile
    const T* ga[LPT];
    const T* gw[LPT];
    int soffa[LPT], soffw[LPT];
    bool cva[LPT], cvw[LPT];                       // K-major: the piece's columns exist
    auto kmajor_off = [](int k, int piece) {       // LDS byte offset of a 16-byte piece of row k
        if constexpr (sizeof(T) == 2) return k * 256 + ((((piece >> 1) ^ ((k & 3) | (((k >> 3) & 1) << 2))) << 5) | ((piece & 1) << 4));
        else return k * 512 + ((((piece >> 2) ^ ((k >> 2) & 3)) << 6) | ((piece & 3) << 4));
    };
#pragma unroll
    for (int i = 0; i < LPT; ++i) {
        int c = tid + NT * i;
        int row = c / NSLOT, slot = c % NSLOT;
        int krow = c / PPR, piece = c % PPR;
        if constexpr (AK) {
            cva[i] = m0 + piece * EPV < p.M;
            ga[i] = (const T*)p.A + (size_t)krow * p.lda + m0 + piece * EPV;
            soffa[i] = kmajor_off(krow, piece);
        } else {
            cva[i] = true;
            ga[i] = (const T*)p.A + (size_t)min(m0 + row, p.M - 1) * p.lda + slot * EPV;
            soffa[i] = row * ROWB + ((slot ^ C::swz(row)) << 4);
        }
        if constexpr (WK) {
            cvw[i] = n0 + piece * EPV < p.N;
            gw[i] = (const T*)p.W + (size_t)krow * p.ldw + n0 + piece * EPV;
            soffw[i] = kmajor_off(krow, piece);
        } else {
            cvw[i] = true;
            gw[i] = (const T*)p.W + (size_t)min(n0 + row, p.N - 1) * p.ldw + slot * EPV;
            soffw[i] = row * ROWB + ((slot ^ C::swz(row)) << 4);
        }
    }
    u32x4 ra[DEPTH][LPT], rw[DEPTH][LPT];
    auto gload = [&](int set, int k0) {          // `set` is a compile-time constant at every call site
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const int krow = (tid + NT * i) / PPR;
            if constexpr (AK) {
                ra[set][i] = u32x4{0u, 0u, 0u, 0u};
                if (cva[i] && k0 + krow < p.k_valid) ra[set][i] = *(const u32x4*)(ga[i] + (size_t)k0 * p.lda);
            } else {
                ra[set][i] = *(const u32x4*)(ga[i] + k0);
            }
            if constexpr (WK) {
                rw[set][i] = u32x4{0u, 0u, 0u, 0u};
                if (cvw[i] && k0 + krow < p.k_valid) rw[set][i] = *(const u32x4*)(gw[i] + (size_t)k0 * p.ldw);
            } else {
                rw[set][i] = *(const u32x4*)(gw[i] + k0);
            }
        }
    };
    auto sstore = [&](int set, int buf) {
        unsigned char* b = smem + buf * 2 * TILEB;
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            *(u32x4*)(b + soffa[i]) = ra[set][i];
            *(u32x4*)(b + TILEB + soffw[i]) = rw[set][i];
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fq = lane >> 4;
    f32x4 accb[4];                                 // GROUPED: sum over k of this wave's A rows (every element the same sum)
#pragma unroll
    for (int i = 0; i < 4; ++i) accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool colsum = GROUPED && p.colsum != nullptr && tn == 0 && wn == 0;
    // 16-bit operands: the MFMA fragments of K-step ks (32 of the slab's 64 k) of the tile in LDS buffer `buf`
    auto frags = [&](int buf, int ks, bf16x8 (&af)[4], bf16x8 (&wf)[4]) {
        const unsigned char* ba = smem + buf * 2 * TILEB;
        const unsigned char* bw = ba + TILEB;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            int ra_ = wm * 64 + t * 16 + fr, rw_ = wn * 64 + t * 16 + fr;
            int slot = ks * 4 + fq;
            // K-major: lanes (fr>>2, fr&3) of a 16-lane group address a 4-row x 16-column block whose rows are
            // k = 32 ks + 8 fq + {0..3} (then +4); the transposed read hands lane fr column fr of that block
            const int kb = ks * 32 + 8 * fq + (fr >> 2), fk = (kb & 3) | (((kb >> 3) & 1) << 2);
            if constexpr (AK) {
                const unsigned char* a0 = ba + kb * 256 + ((((wm * 64 + t * 16) >> 4) ^ fk) << 5) + 8 * (fr & 3);
                bf16x4 lo = ds_read_tr16_b64(a0);
                bf16x4 hi = ds_read_tr16_b64(a0 + 4 * 256);
                af[t] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            } else {
                af[t] = *(const bf16x8*)(ba + ra_ * ROWB + ((slot ^ C::swz(ra_)) << 4));
            }
            if constexpr (WK) {
                const unsigned char* w0 = bw + kb * 256 + ((((wn * 64 + t * 16) >> 4) ^ fk) << 5) + 8 * (fr & 3);
                bf16x4 lo = ds_read_tr16_b64(w0);
                bf16x4 hi = ds_read_tr16_b64(w0 + 4 * 256);
                wf[t] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            } else {
                wf[t] = *(const bf16x8*)(bw + rw_ * ROWB + ((slot ^ C::swz(rw_)) << 4));
            }
        }
    };
    auto mma = [&](const bf16x8 (&af)[4], const bf16x8 (&wf)[4]) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                acc[mt][nt] = STDIFF_MFMA_16x16x32_H16(wf[nt], af[mt], acc[mt][nt], 0, 0, 0);
        if constexpr (GROUPED) {
            if (colsum) {
                const h16 one = (h16)1.0f;
                const bf16x8 ones = {one, one, one, one, one, one, one, one};
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) accb[mt] = STDIFF_MFMA_16x16x32_H16(ones, af[mt], accb[mt], 0, 0, 0);
            }
        }
    };
    auto compute = [&](int buf) {
        const unsigned char* ba = smem + buf * 2 * TILEB;
        const unsigned char* bw = ba + TILEB;
        (void)bw;
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 af[4], wf[4];
                frags(buf, ks, af, wf);
                mma(af, wf);
            }
        } else {
            f32x4 af[4], wf[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                int ra_ = wm * 64 + t * 16 + fr, rw_ = wn * 64 + t * 16 + fr;
                if constexpr (AK) {            // element kk of lane (fr, fq) is k = 4 fq + kk: row 4 fq + kk, column ra_
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk)
                        af[t][kk] = *(const float*)(ba + (4 * fq + kk) * 512 + ((((wm * 64 + t * 16) >> 4) ^ fq) << 6) + fr * 4);
                } else {
                    af[t] = *(const f32x4*)(ba + ra_ * ROWB + ((fq ^ C::swz(ra_)) << 4));
                }
                if constexpr (WK) {
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk)
                        wf[t][kk] = *(const float*)(bw + (4 * fq + kk) * 512 + ((((wn * 64 + t * 16) >> 4) ^ fq) << 6) + fr * 4);
                } else {
                    wf[t] = *(const f32x4*)(bw + rw_ * ROWB + ((fq ^ C::swz(rw_)) << 4));
                }
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[nt][kk], af[mt][kk], acc[mt][nt], 0, 0, 0);
            if constexpr (GROUPED) {
                if (colsum) {
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt)
                            accb[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, af[mt][kk], accb[mt], 0, 0, 0);
                }
            }
        }
    };

    const int kb = split * p.nk_per, nk = min(p.K / BK - kb, p.nk_per);     // this block's K-slabs
    if constexpr (DEPTH == 1) {
        gload(0, kb * BK);
        sstore(0, 0);
        __syncthreads();
        for (int k = 0; k < nk; ++k) {
            if (k + 1 < nk) gload(0, (kb + k + 1) * BK);
            compute(k & 1);
            if (k + 1 < nk) sstore(0, (k + 1) & 1);
            __syncthreads();
        }
    } else {
        // Register set j % DEPTH carries slab j.  Every load and store below is unconditional (slab indices clamped to
        // the last one: a few reloads at the tail whose LDS copies nobody reads) so that the compiler's counted waits
        // see one straight-line load stream: the store of slab j + 1 waits with (DEPTH - 1) slabs still in flight.
        const int last = kb + nk - 1;
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) gload(d, min(kb + d, last) * BK);
        sstore(0, 0);
        __syncthreads();
        int j = 0;
        if constexpr (sizeof(T) == 2) {
            // One wave per SIMD: nothing else hides the LDS latency of the fragment reads, so they are software-pipelined
            // across the barrier -- the second K-step's fragments are read under the first one's 16 MFMAs, and the NEXT
            // slab's first fragments (from the buffer the barrier has just published) under the second one's.
            bf16x8 af0[4], wf0[4], af1[4], wf1[4];
            frags(0, 0, af0, wf0);
            // sched_barrier(0): the machine scheduler would otherwise hoist the MFMAs that consume the reads just issued
            // above the ones whose operands are already in registers, exposing the LDS latency again
            // The LDS store of the next slab (loaded two steps ago) is issued BEFORE the first 16 MFMAs, so that it
            // drains under them instead of in front of the barrier.
            auto step = [&](int buf, int next_set, auto prefetch) {
                frags(buf, 1, af1, wf1);
                prefetch();
                sstore(next_set, buf ^ 1);
                __builtin_amdgcn_sched_barrier(0);
                mma(af0, wf0);
                __builtin_amdgcn_sched_barrier(0);
            };
            auto step2 = [&](int buf) {
                __syncthreads();
                frags(buf ^ 1, 0, af0, wf0);
                __builtin_amdgcn_sched_barrier(0);
                mma(af1, wf1);
                __builtin_amdgcn_sched_barrier(0);
            };
            for (; j + DEPTH <= nk; j += DEPTH) {
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) {
                    const int buf = (j + d) & 1;
                    step(buf, (d + 1) % DEPTH, [&] { gload(d, min(kb + j + d + DEPTH, last) * BK); });
                    step2(buf);
                }
            }
#pragma unroll
            for (int d = 0; d < DEPTH - 1; ++d) {
                if (j + d < nk) {
                    const int buf = (j + d) & 1;
                    step(buf, (d + 1) % DEPTH, [] {});
                    step2(buf);
                }
            }
        } else {
            for (; j + DEPTH <= nk; j += DEPTH) {
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) {
                    gload(d, min(kb + j + d + DEPTH, last) * BK);
                    compute((j + d) & 1);
                    sstore((d + 1) % DEPTH, (j + d + 1) & 1);
                    __syncthreads();
                }
            }
#pragma unroll
            for (int d = 0; d < DEPTH - 1; ++d) {
                if (j + d < nk) {
                    compute((j + d) & 1);
                    sstore((d + 1) % DEPTH, (j + d + 1) & 1);
                    __syncthreads();
                }
            }
        }
    }

    if constexpr (GROUPED) {
        if (p.sumsq) {
            // the tile's share of the squared gradient norm, from the accumulators: wave sums into the 16 bytes behind the
            // tile buffers, one plain store per tile after the barrier below (atomics on one cell were tried first:
            // 18 000 of them per launch at H = 2048 serialise at ~6 ns each and made the launch 160 -> 269 us)
            float ss = 0.f;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int gm = m0 + wm * 64 + mt * 16 + fr;
                float rs = 0.f;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const int gn = n0 + wn * 64 + nt * 16 + fq * 4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) rs += gn + e < p.N ? acc[mt][nt][e] * acc[mt][nt][e] : 0.f;
                }
                if (colsum && fq == 0) rs += accb[mt][0] * accb[mt][0];
                ss += gm < p.M ? rs : 0.f;
            }
            ss = wave_sum(ss);
            if (lane == 0) ((float*)(smem + smem128<T>()))[wid] = ss;
        }
    }
    // ---- accumulators -> fp32 C image in LDS (lane holds 4 consecutive n of one m)
    float* cl = (float*)smem;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            int m = wm * 64 + mt * 16 + fr, n = wn * 64 + nt * 16 + fq * 4;
            *(f32x4*)(cl + m * CLD + n) = acc[mt][nt];
        }
    __syncthreads();
    if constexpr (GROUPED) {
        if (colsum && fq == 0) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int gm = m0 + wm * 64 + mt * 16 + fr;
                if (gm < p.M) p.colsum[gm] = accb[mt][0];
            }
        }
        if (p.sumsq && tid == 0) {
            const float* sq = (const float*)(smem + smem128<T>());
            *p.sumsq = (sq[0] + sq[1]) + (sq[2] + sq[3]);
        }
        epi_rows128<STDIFF_EPI_BIAS, T>(p, cl, tid, m0, n0);
        return;
    }
    if (p.ksplit > 1) {                 // partial product -> workspace slice; splitk_reduce_kernel finishes the job
        float* w = p.ws + (size_t)split * p.M * p.ldws;
#pragma unroll 1
        for (int i = 0; i < 16; ++i) {
            const int r = (tid >> 5) + 8 * i, c4 = (tid & 31) * 4;
            const int gm = m0 + r, gn = n0 + c4;
            if (gm < p.M && gn < p.ldws) *(f32x4*)(w + (size_t)gm * p.ldws + gn) = *(const f32x4*)(cl + r * CLD + c4);
        }
        return;
    }
    if (p.vec) {
#define STDIFF_CALL128(E) epi_rows128<E, T>(p, cl, tid, m0, n0)
        STDIFF_EPI_DISPATCH(STDIFF_CALL128)
        return;
    }
#pragma unroll 1
    for (int i = 0; i < 16; ++i) {
        const int r = (tid >> 4) + 16 * (i >> 1), c4 = (tid & 15) * 8 + 4 * (i & 1);
        const int gm = m0 + r, gn = n0 + c4;
        if (gm < p.M && gn < p.N) epilogue4_slow<T>(p, gm, gn, *(const f32x4*)(cl + r * CLD + c4));
    }
}

template <typename T, bool AK, bool WK, int DEPTH = 1>
__global__ __launch_bounds__(NT, DEPTH == 1 ? 2 : 1) void gemm_nt_kernel(Params p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[smem128<T>()];
    gemm_nt_body<T, AK, WK, DEPTH, false>(p, blockIdx.x, smem);
}

// Several weight-gradient products of one batch in ONE launch: gw_i[M_i, N_i] = dY_i^T X_i, gb_i = colsum(dY_i), all
// over the same K = cells, both operands read K-major as they lie.  A DiT block's five products are 1152 tiles at
// H = 1024 -- four rounds of the chip with two blocks per CU, so no K split, no reduce pass and no separate column-sum
// launch (15 launches become 1).  Problem i owns blocks [end_{i-1}, end_i), each range rounded up to a multiple of 8 so
// that the XCD-aware tile walk of every problem starts on XCD 0; the padding blocks exit at once.
constexpr int DW_MAX = 8;
struct DwProblem {
    int M, N;
    const void* A; int lda;
    const void* W; int ldw;
    float* C; int ldc;
    float* colsum;
    int tiles_m, tiles_n, blk_end;
};
struct DwGroup {
    int n, K, k_valid, total;
    float* sumsq;
    DwProblem pr[DW_MAX];
};
template <typename T>
__global__ __launch_bounds__(NT, 2) void gemm_dw_group_kernel(DwGroup g) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[smem128<T>() + 16];      // + four wave sums (g.sumsq)
    // the grid may be smaller than the tile count (one resident block per CU, so that the input-gradient chain that
    // runs beside this launch keeps a slot on every CU): each block then walks tiles bid, bid + grid, ...
#pragma unroll 1
    for (int bid = blockIdx.x; bid < g.total; bid += gridDim.x) {
    int i = 0, start = 0;
#pragma unroll 1
    while (i + 1 < g.n && bid >= g.pr[i].blk_end) { start = g.pr[i].blk_end; ++i; }
    const DwProblem& q = g.pr[i];
    const int local = bid - start;
    if (local >= q.tiles_m * q.tiles_n) {          // padding block: its slots of the squared-norm parts are zero
        if (g.sumsq && threadIdx.x == 0) g.sumsq[bid] = 0.f;
        continue;
    }
    Params p;
    p.M = q.M; p.N = q.N; p.K = g.K;
    p.A = q.A; p.lda = q.lda; p.W = q.W; p.ldw = q.ldw; p.bias = nullptr;
    p.C = q.C; p.ldc = q.ldc; p.c_bf16 = 0;
    p.epi = STDIFF_EPI_BIAS; p.gate = nullptr; p.ldgate = 0; p.res = nullptr; p.ldres = 0;
    p.C2 = nullptr; p.ldc2 = 0; p.vec = 1;
    p.tiles_m = q.tiles_m; p.tiles_n = q.tiles_n; p.k_valid = g.k_valid;
    p.ksplit = 1; p.nk_per = g.K / Cfg<T>::BK; p.ws = nullptr; p.ldws = 0;
    p.t_full = p.units = 0; p.ln_parts = nullptr; p.ln_nparts = 0; p.stats_out = nullptr;
    p.colsum = q.colsum; p.sumsq = g.sumsq ? g.sumsq + bid : nullptr;
    gemm_nt_body<T, true, true, 1, true>(p, local, smem);
    __syncthreads();                       // the C image of this tile is read before the next tile's staging overwrites it
    }
}

// sum of the split-K partial products in slice order (deterministic) + the full epilogue; one thread = 4 columns
template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(Params p) {
    const int per_row = p.ldws >> 2;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= p.M * per_row) return;
    const int gm = idx / per_row, gn = (idx - gm * per_row) * 4;
    const float* w = p.ws + (size_t)gm * p.ldws + gn;
    f32x4 v = *(const f32x4*)w;
    for (int s = 1; s < p.ksplit; ++s) v += *(const f32x4*)(w + (size_t)s * p.M * p.ldws);
    epilogue4_slow<T>(p, gm, gn, v);
}

// ================================================================ bf16, 256 x 256 tile, 4 phases per K-slab, counted DMA waits
// One workgroup per CU, 8 waves as 2 (M) x 4 (N).  The tile is cut into half-tiles A0/A1 (rows 0-127 / 128-255 of the
// block) and B0/B1 (columns); wave (wr, wc) owns rows {128*mh + 64*wr + 0..63} and columns {128*nh + 32*wc + 0..31}
// for mh, nh in {0,1}, so in phase (mh, nh) EVERY wave reads half-tiles A_mh and B_nh only:
//     Q1 = (A0,B0)   Q2 = (A0,B1)   Q3 = (A1,B1)   Q4 = (A1,B0)        16 MFMAs (64 x 32 x K64) per wave per phase
// A half-tile therefore retires early (A0 after Q2, B1 after Q3, A1/B0 after Q4) and is refilled by LDS-DMA
// (global_load_lds, 16 B per lane, 1 KiB = 8 rows per wave-instruction, no VGPR round trip) two phases later, ONE
// half-tile (2 instructions per lane) per phase:
//     t.Q1: B1(t+1)   t.Q2: B0(t+1)   t.Q3: A1(t+1)   t.Q4: A0(t+2)
// so up to 3 half-tiles stay in flight across the barriers.  The LDS image is lane-linear, so the XOR slot swizzle is
// applied to the per-lane SOURCE address and to the fragment read (cdna guide 5.4 rule 21).  Waits are counted, never
// vmcnt(0) in steady state: vmcnt(6) in Q2 (retires A1(t), read in Q3) and vmcnt(4) in Q4 (retires A0,B1,B0 of t+1,
// read from the next Q1 on); each wait sits before that phase's barrier and the data is read one phase later.
// Waves 4-7 (wr = 1) run one barrier behind waves 0-3, so on every SIMD one wave is in its MFMA cluster while its
// partner reads fragments / issues DMA.  Two barriers per phase; restage >= 2 phases after the last read, read >= 1
// phase after the wait.  Measured per 256x256x64 slab: 1.46 us (1.47 PFLOP/s chip-wide at K = 1024).
#define STDIFF_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define STDIFF_BARRIER()               \
    do {                               \
        __builtin_amdgcn_s_barrier();  \
        asm volatile("" ::: "memory"); \
    } while (0)

#define STDIFF_LGKMCNT0() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")

// 64 rows of the swizzled fp32 C quarter image `cl` [64][256] -> global.  Image row i = 32*wr + 16*j + fr holds block
// row 128*mh + 64*wr + 16*(2*mp + j) + fr (quarter q = 2*mh + mp); wave w stores image rows w, w+8, ..: every
// wave-instruction covers one whole 256-column row (1 KiB fp32 / 512 B bf16), the row index is wave-uniform (scalar
// bounds check), and the 4 LDS reads of a batch are issued together, ahead of the checks.  Output type and the C2
// copy are compile-time, as the epilogue kind is.
template <int EPI, bool CBF, bool HASC2>
__device__ __forceinline__ void epi_quarter256(const Params& p, const float* cl, int w, int cq, int m0, int n0, int q) {
    const int gn = n0 + cq * 4;
    if (gn >= p.N) return;
    const f32x4 bias4 = p.bias ? *(const f32x4*)(p.bias + gn) : f32x4{0.f, 0.f, 0.f, 0.f};
    const int rbase = m0 + (q >> 1) * 128 + (q & 1) * 32;
#pragma unroll 1
    for (int pb = 0; pb < 2; ++pb) {
        f32x4 g[4], r[4], v[4];
        int gm[4];
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int i = w + 8 * (pb * 4 + ps);
            gm[ps] = rbase + (i >> 5) * 64 + ((i >> 4) & 1) * 16 + (i & 15);
            epi_load<EPI, bf16_t>(p, min(gm[ps], p.M - 1), gn, g[ps], r[ps]);
        }
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int i = w + 8 * (pb * 4 + ps);
            v[ps] = *(const f32x4*)(cl + i * 256 + ((cq ^ (i & 15)) << 2));
        }
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) asm volatile("" : "+v"(v[ps]));
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            if (gm[ps] < p.M) {
                const f32x4 pre = v[ps] + bias4;
                const f32x4 o = epi_math<EPI, CBF>(pre, g[ps], r[ps]);
                // streaming stores: C is far larger than L2 and is next read by another kernel
                if constexpr (CBF) {
                    u32x2 rr;
                    rr[0] = pack2bf(o[0], o[1]);
                    rr[1] = pack2bf(o[2], o[3]);
                    __builtin_nontemporal_store(rr, (u32x2*)((bf16_t*)p.C + (size_t)gm[ps] * p.ldc + gn));
                } else __builtin_nontemporal_store(o, (f32x4*)((float*)p.C + (size_t)gm[ps] * p.ldc + gn));
                if constexpr (HASC2) st4((bf16_t*)p.C2 + (size_t)gm[ps] * p.ldc2 + gn, pre);
            }
        }
    }
}

// LDS accesses of the gate-residual tile epilogue are written as asm so that the compiler's conservative
// s_waitcnt vmcnt(0) (LDS access while an LDS-DMA may be in flight) does not drain the prefetched side loads.
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}
__device__ __forceinline__ void lds_write16(unsigned addr, f32x4 v) {
    asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ f32x4 lds_read16(unsigned addr) {
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}

// Whole-tile epilogue of the two-phase kernel for out = res + gate * (acc + bias) (attn.proj / mlp.fc2: the fp32
// residual stream).  Same quarter images as epi_quarter256, but the side inputs (gate: 8 B, res: 16 B per lane and
// row) of batch b + 2 are requested BEFORE batch b is read from the image, converted and stored: their HBM latency
// (8 batches per tile, ~1.5 us each when exposed) runs under the two previous batches and under the put + barrier of
// the next quarter (SLOTS = 3: two batches = 96 KiB per CU in flight; SLOTS = 2: one).  Measured at a 131 072-row slab:
// fc2 968 -> 955 us, proj unchanged (its 2.4 GB per launch already move at 5.4 TB/s: HBM-bound as a whole).  Fully
// unrolled so that the register sets are static.
template <bool CBF, bool STATS = false, int SLOTS = 2>
__device__ __forceinline__ void epi_tile256_gate_res(const Params& p, const f32x4 (&acc)[2][2][4][2], unsigned cla,
                                                     int wid, int lane, int m0, int n0, int nq) {
    asm volatile("" : "+v"(lane));       // per-lane offsets are recomputed per tile, not kept live across the K loop
    const int wr = wid >> 2, wc = wid & 3, fr = lane & 15, fq = lane >> 4;
    const int gn = n0 + lane * 4;
    const bool colok = gn < p.N;
    const int gnc = min(gn, p.N - 4);
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) bias4 = *(const f32x4*)(p.bias + gnc);
    u32x2 graw[SLOTS][4];            // SLOTS - 1 batches ahead (3 where the kernel's register budget allows it)
    f32x4 r[SLOTS][4];
    auto row_of = [&](int b, int ps) {                   // batch b = 2 * quarter + pb
        const int q = b >> 1, i = wid + 8 * ((b & 1) * 4 + ps);
        return m0 + (q >> 1) * 128 + (q & 1) * 32 + (i >> 5) * 64 + ((i >> 4) & 1) * 16 + (i & 15);
    };
    auto issue = [&](int b, int slot) {
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
            const int gm = min(row_of(b, ps), p.M - 1);
            graw[slot][ps] = *(const u32x2*)((const bf16_t*)p.gate + (size_t)gm * p.ldgate + gnc);
            r[slot][ps] = *(const f32x4*)(p.res + (size_t)gm * p.ldres + gnc);
        }
    };
    issue(0, 0);
    if constexpr (SLOTS > 2) issue(1, 1);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (q >= nq) break;                              // half unit: two quarters
        if (q) STDIFF_BARRIER();                         // image reads of the previous quarter are done
        const int mh = q >> 1, mp = q & 1;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const int row = wr * 32 + j * 16 + fr;
                    const int chunk = nh * 32 + wc * 8 + nt * 4 + fq;
                    lds_write16(cla + (unsigned)(row * 256 + ((chunk ^ fr) << 2)) * 4u, acc[mh][nh][2 * mp + j][nt]);
                }
        STDIFF_LGKMCNT0();
        STDIFF_BARRIER();
#pragma unroll
        for (int pb = 0; pb < 2; ++pb) {
            const int b = 2 * q + pb, slot = b % SLOTS;
            if (b + SLOTS - 1 < 8) issue(b + SLOTS - 1, (b + SLOTS - 1) % SLOTS);
            f32x4 v[4];
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const int i = wid + 8 * (pb * 4 + ps);
                v[ps] = lds_read16(cla + (unsigned)(i * 256 + ((lane ^ (i & 15)) << 2)) * 4u);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3])::"memory");
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                f32x4 g;
                g[0] = bf_lo(graw[slot][ps][0]);
                g[1] = bf_hi(graw[slot][ps][0]);
                g[2] = bf_lo(graw[slot][ps][1]);
                g[3] = bf_hi(graw[slot][ps][1]);
                v[ps] = r[slot][ps] + g * (v[ps] + bias4);
                asm volatile("" : "+v"(v[ps]));          // keep the math out of the predicated blocks
            }
            if constexpr (STATS) {
                // the 64 lanes of this wave hold one whole 256-column row of the new residual stream per `ps`: its
                // partial moments (mean_p, M2_p) go to the next LayerNorm (stdiff_gemm_args.stats_out)
#pragma unroll
                for (int ps = 0; ps < 4; ++ps) {
                    const float mp = wave_sum((v[ps][0] + v[ps][1]) + (v[ps][2] + v[ps][3])) * (1.0f / 256.0f);
                    float q2 = 0.f;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { const float d = v[ps][j] - mp; q2 += d * d; }
                    q2 = wave_sum(q2);
                    const int gm = row_of(b, ps);
                    if (lane == 0 && gm < p.M) {
                        float* o = p.stats_out + ((size_t)gm * p.tiles_n + (n0 >> 8)) * 2;
                        o[0] = mp; o[1] = q2;
                    }
                }
            }
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const int gm = row_of(b, ps);
                if (gm < p.M && colok) {
                    if constexpr (CBF) {
                        u32x2 rr;
                        rr[0] = pack2bf(v[ps][0], v[ps][1]);
                        rr[1] = pack2bf(v[ps][2], v[ps][3]);
                        __builtin_nontemporal_store(rr, (u32x2*)((bf16_t*)p.C + (size_t)gm * p.ldc + gn));
                    } else __builtin_nontemporal_store(v[ps], (f32x4*)((float*)p.C + (size_t)gm * p.ldc + gn));
                }
            }
        }
    }
}

// Whole-tile epilogue of the two-phase kernel for bf16 outputs without side inputs (bias / SiLU / GELU: adaLN, qkv,
// fc1, the final adaLN): the epilogue arithmetic runs on the accumulator registers, the RESULT is staged as bf16 --
// one 128-row x 256-column image (64 KiB) per M half, so a tile takes 2 put/barrier/copy rounds instead of 4 and moves
// half the LDS bytes of the fp32 quarter images -- and leaves as 16-byte stores, 32 lanes to a 512-byte row.
// Image: row r (512 B) holds 64 8-byte chunks; chunk c of row r sits at c ^ ((r & 7) << 2) ^ (((r >> 3) & 1) << 1):
// the 32 lanes of a half wave (16 rows x 2 neighbouring chunks) then write 32 distinct bank pairs, and a 16-byte
// read unit (chunks 2p, 2p + 1) stays adjacent because bit 0 is never flipped.
template <int EPI>
__device__ __forceinline__ void epi_tile256_bf16(const Params& p, const f32x4 (&acc)[2][2][4][2], unsigned char* img,
                                                 int tid, int wid, int lane, int m0, int n0, int nq) {
    asm volatile("" : "+v"(lane), "+v"(tid));   // per-lane offsets are recomputed per tile, not kept live across the K loop
    const int wr = wid >> 2, wc = wid & 3, fr = lane & 15, fq = lane >> 4;
    f32x4 b4[2][2];
#pragma unroll
    for (int nh = 0; nh < 2; ++nh)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int gn = n0 + nh * 128 + wc * 32 + nt * 16 + fq * 4;
            b4[nh][nt] = (p.bias && gn < p.N) ? *(const f32x4*)(p.bias + gn) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    const int sw = ((fr & 7) << 2) ^ ((fr >> 3) << 1);
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int mh = 0; mh < 2; ++mh) {
        if (2 * mh >= nq) break;                         // half unit: rows of acc[0] only
        if (mh) STDIFF_BARRIER();                        // image reads of the first half are done
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const f32x4 o = epi_math<EPI, true>(acc[mh][nh][mt][nt] + b4[nh][nt], zero, zero);
                    u32x2 rr;
                    rr[0] = pack2bf(o[0], o[1]);
                    rr[1] = pack2bf(o[2], o[3]);
                    const int r = wr * 64 + mt * 16 + fr, c = nh * 32 + wc * 8 + nt * 4 + fq;
                    *(u32x2*)(img + r * 512 + ((c ^ sw) << 3)) = rr;
                }
        STDIFF_LGKMCNT0();
        STDIFF_BARRIER();
        const int pair = tid & 31, gn = n0 + pair * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int row = (tid >> 5) + 16 * k, gm = m0 + mh * 128 + row;
            const int rs = ((row & 7) << 2) ^ (((row >> 3) & 1) << 1);
            const u32x4 v = *(const u32x4*)(img + row * 512 + (((2 * pair) ^ rs) << 3));
            if (gm < p.M && gn < p.N) __builtin_nontemporal_store(v, (u32x4*)((bf16_t*)p.C + (size_t)gm * p.ldc + gn));
        }
        STDIFF_LGKMCNT0();
    }
}

// Whole-tile epilogue of the two-phase kernel for the adaLN projection fused with the LayerNorm + modulate it feeds
// (STDIFF_EPI_LN_MOD).  The tile's 256 weight rows are [128 shift | 128 scale] of the same 128 features, i.e. the B0 / B1
// column halves: acc[mh][0] holds shift and acc[mh][1] scale for the same (row, feature) in the same lane.  The lane
// rounds both to the 16-bit format (what the unfused path stores and re-reads), loads its 4 features of the fp32
// residual row and the row's statistics, applies ln_mod_apply and stages the 16-bit result -- one 128-row x 128-column
// image (32 KiB) per M half -- for 16-byte stores, 16 lanes to a 256-byte row.
// Image: row r (256 B) holds 32 8-byte chunks; chunk c of row r sits at c ^ ((r & 15) << 1): the 16 lanes of a
// ds_write_b64 group (16 rows, one chunk) hit 16 distinct bank pairs, and bit 0 is never flipped, so a 16-byte read
// unit stays adjacent.
// HBM latency is what this epilogue costs (128 KiB of residual rows per tile, one workgroup per CU), so every request
// is in flight as early as it can be: the residual loads of the first M half (8 x 16 B per lane) go out first, thread
// r < 256 combines row r's partial moments ONCE per tile into (mean, rstd) in LDS (2 KiB behind the image) while they
// fly, and the second half's residual loads are issued before the first half is converted.  LDS accesses are asm for
// the same reason as in epi_tile256_gate_res.
#ifdef STDIFF_GEMM_STAMPS
__device__ unsigned long long g_stamps[256 * 8 * 8];      // [block][tile][0..3 realtime (100 MHz), 4..7 shader clock]
#define STDIFF_STAMP(k) do { if (tid == 0 && blockIdx.x < 256 && iter < 8) { g_stamps[(blockIdx.x * 8 + iter) * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); g_stamps[(blockIdx.x * 8 + iter) * 8 + 4 + (k)] = __builtin_amdgcn_s_memtime(); } } while (0)
// sub-stamps of the LN_MOD epilogue go to the shader-clock slots (realtime): 4 statistics ready, 5 first half staged,
// 6 first half stored, 7 second half staged
#define STDIFF_SUBSTAMP(k) do { if (tid == 0 && blockIdx.x < 256 && dbg_iter < 8) g_stamps[(blockIdx.x * 8 + dbg_iter) * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define STDIFF_SUBSTAMP(k) do {} while (0)
#define STDIFF_STAMP(k) do {} while (0)
#endif
__device__ __forceinline__ void lds_write8(unsigned addr, u32x2 v) {
    asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ u32x2 lds_read8(unsigned addr) {
    u32x2 v;
    asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}
__device__ __forceinline__ void epi_tile256_lnmod(const Params& p, const f32x4 (&acc)[2][2][4][2], unsigned imga,
                                                  int tid, int wid, int lane, int m0, int n0, int nq, int dbg_iter = 0) {
    asm volatile("" : "+v"(lane), "+v"(tid));   // per-lane offsets are recomputed per tile, not kept live across the K loop
    const int wr = wid >> 2, wc = wid & 3, fr = lane & 15, fq = lane >> 4;
    const int c0 = n0 >> 1;                      // first output column of this tile
    const unsigned sta = imga + 32768u;          // (mean, rstd) of the tile's 256 rows
    f32x4 x[2][4][2];
    auto issue = [&](int mh, int slot) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int gm = min(m0 + mh * 128 + wr * 64 + mt * 16 + fr, p.M - 1);
            const float* xr = p.res + (size_t)gm * p.ldres + c0 + wc * 32 + fq * 4;
            x[slot][mt][0] = *(const f32x4*)xr;
            x[slot][mt][1] = *(const f32x4*)(xr + 16);
        }
    };
    issue(0, 0);
    if (tid < 256) {
        // row statistics from the partial moments (fixed order; one part: exactly stdiff_ln_modulate's)
        const int np = p.ln_nparts;
        const float* pp = p.ln_parts + (size_t)min(m0 + tid, p.M - 1) * np * 2;
        float pm[8], pq[8];
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (q < np) {
                const f32x2_cv t = *(const f32x2_cv*)(pp + 2 * q);
                pm[q] = t[0];
                pq[q] = t[1];
            } else pm[q] = pq[q] = 0.f;
        float mean = pm[0], m2 = pq[0];
        if (np > 1) {
#pragma unroll
            for (int q = 1; q < 8; ++q)
                if (q < np) { mean += pm[q]; m2 += pq[q]; }
            mean /= (float)np;
            float dev = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (q < np) { const float d = pm[q] - mean; dev += d * d; }
            m2 += dev * (float)((p.N >> 1) / np);
        }
        const float rstd = rsqrtf(m2 / (float)(p.N >> 1) + 1e-6f);
        u32x2 st;
        st[0] = __float_as_uint(mean);
        st[1] = __float_as_uint(rstd);
        lds_write8(sta + (unsigned)tid * 8u, st);
    }
    f32x4 bsh[2], bsc[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int gn = n0 + wc * 32 + nt * 16 + fq * 4;
        bsh[nt] = p.bias ? *(const f32x4*)(p.bias + gn) : f32x4{0.f, 0.f, 0.f, 0.f};
        bsc[nt] = p.bias ? *(const f32x4*)(p.bias + gn + 128) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // shift and 1 + scale, rounded through the 16-bit format, for both halves: arithmetic that needs neither the
    // residual rows nor the statistics runs while those loads are in flight
    f32x4 shv[2][4][2], opv[2][4][2];
#pragma unroll
    for (int mh = 0; mh < 2; ++mh)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const f32x4 sh = acc[mh][0][mt][nt] + bsh[nt], sc = acc[mh][1][mt][nt] + bsc[nt];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    shv[mh][mt][nt][j] = bf2f(f2bf(sh[j]));
                    opv[mh][mt][nt][j] = 1.0f + bf2f(f2bf(sc[j]));
                }
            }
    STDIFF_LGKMCNT0();
    STDIFF_BARRIER();
    STDIFF_SUBSTAMP(4);
#pragma unroll
    for (int mh = 0; mh < 2; ++mh) {
        if (2 * mh >= nq) break;                         // half unit: rows of acc[0] only
        if (mh) STDIFF_BARRIER();                        // image reads of the first half are done
        if (mh == 0 && nq > 2) issue(1, 1);
        u32x2 st[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) st[mt] = lds_read8(sta + (unsigned)(mh * 128 + wr * 64 + mt * 16 + fr) * 8u);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(st[0]), "+v"(st[1]), "+v"(st[2]), "+v"(st[3])::"memory");
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int r = wr * 64 + mt * 16 + fr;
            const float mean = __uint_as_float(st[mt][0]), rstd = __uint_as_float(st[mt][1]);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const f32x4 xv = x[mh][mt][nt];
                f32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = ln_mod_apply_pre(xv[j], mean, rstd, opv[mh][mt][nt][j], shv[mh][mt][nt][j]);
                u32x2 rr;
                rr[0] = pack2bf(o[0], o[1]);
                rr[1] = pack2bf(o[2], o[3]);
                const int c = wc * 8 + nt * 4 + fq;
                lds_write8(imga + (unsigned)(r * 256 + ((c ^ ((r & 15) << 1)) << 3)), rr);
            }
        }
        STDIFF_LGKMCNT0();
        STDIFF_BARRIER();
        if (mh) STDIFF_SUBSTAMP(6);
        const int pair = tid & 15, gn = c0 + pair * 8;
        f32x4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int row = (tid >> 4) + 32 * k;
            v[k] = lds_read16(imga + (unsigned)(row * 256 + (((2 * pair) ^ ((row & 15) << 1)) << 3)));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3])::"memory");
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int gm = m0 + mh * 128 + (tid >> 4) + 32 * k;
            if (gm < p.M) __builtin_nontemporal_store(v[k], (f32x4*)((bf16_t*)p.C + (size_t)gm * p.ldc + gn));
        }
        if (!mh) STDIFF_SUBSTAMP(5);
    }
}

// PERSISTENT: the grid is min(tiles, CUs) workgroups and each walks tiles b, b + grid, ...  The LDS holds 9 half-tile
// slots (144 KiB): [buf0: A0 A1 B0 B1][buf1: A0 A1 B0 B1][extra]; the epilogue's fp32 C image (64 rows at a time) lives
// in the LAST FOUR slots, so the next tile's prologue (slab 0 into buf0 and A0 of slab 1 into buf1.A0) is issued by
// LDS-DMA BEFORE the epilogue and lands while the block converts and stores its C tile: block launch, prologue
// latency and most of the store drain disappear from the per-tile cost.  The epilogue uses raw barriers + lgkmcnt
// waits (a __syncthreads would wait for the DMA and the global stores as well).
__global__ __launch_bounds__(512, 1) void gemm256p_kernel(Params p) {
    constexpr int ROWB = 128, HALF = 128 * ROWB;          // one half-tile = 128 rows x 64 bf16 = 16 KiB
    __shared__ __attribute__((aligned(16))) unsigned char smem[9 * HALF];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;
    const int fr = lane & 15, fq = lane >> 4;
    const int ntiles = p.tiles_m * p.tiles_n;
    const int nk = p.K / 64;

    // DMA sources: half-tile h, piece q = 2*wid + i (8 rows): lane -> row 8q + (lane>>3), LDS slot lane&7
    const bf16_t* src[4][2];       // [A0 A1 B0 B1][piece]
    auto set_tile = [&](int tile, int& m0, int& n0) {
        int tm, tn;
        tile_of(tile, ntiles, p.tiles_m, p.tiles_n, tm, tn);
        m0 = tm * 256; n0 = tn * 256;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = (2 * wid + i) * 8 + (lane >> 3);
                const int sw = ((lane & 7) ^ (row & 7)) << 3;
                src[h][i] = (const bf16_t*)p.A + (size_t)min(m0 + h * 128 + row, p.M - 1) * p.lda + sw;
                src[2 + h][i] = (const bf16_t*)p.W + (size_t)min(n0 + h * 128 + row, p.N - 1) * p.ldw + sw;
            }
    };
    auto dma = [&](int buf, int slot, int kt) {       // half-tile `slot` of K-slab kt into buffer buf
        unsigned char* dst = smem + (buf * 4 + slot) * HALF + (2 * wid) * 1024;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[slot][i] + kt * 64),
                                             (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
    };
    auto prologue = [&]() {                           // slab 0 complete + A0 of slab 1
        dma(0, 0, 0); dma(0, 2, 0); dma(0, 3, 0); dma(0, 1, 0);
        if (nk > 1) dma(1, 0, 1);
    };

    // fragment read offsets inside a half-tile (row & 7 == fr & 7 for every fragment row)
    int offA[2], offB[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int sl = ((ks * 4 + fq) ^ (fr & 7)) << 4;
        offA[ks] = (wr * 64 + fr) * ROWB + sl;
        offB[ks] = (wc * 32 + fr) * ROWB + sl;
    }
    bf16x8 af[4][2], bfr[2][2];
    auto loadA = [&](int buf, int mh) {
        const unsigned char* b = smem + (buf * 4 + mh) * HALF;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) af[mt][ks] = *(const bf16x8*)(b + offA[ks] + mt * 16 * ROWB);
    };
    auto loadB = [&](int buf, int nh) {
        const unsigned char* b = smem + (buf * 4 + 2 + nh) * HALF;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) bfr[nt][ks] = *(const bf16x8*)(b + offB[ks] + nt * 16 * ROWB);
    };

    f32x4 acc[2][2][4][2];

#define STDIFF_MMA(MH, NH)                                                                                    \
    do {                                                                                                      \
        __builtin_amdgcn_s_setprio(1);                                                                        \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                      \
        _Pragma("unroll") for (int mt = 0; mt < 4; ++mt)                                                      \
        _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                                      \
            acc[MH][NH][mt][nt] =                                                                             \
                STDIFF_MFMA_16x16x32_H16(bfr[nt][ks], af[mt][ks], acc[MH][NH][mt][nt], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                                        \
    } while (0)

    // C quarter image (slots 5..8), XOR-swizzled 16-byte chunks
    float* cl = (float*)(smem + 5 * HALF);
    auto put = [&](auto MH, auto MP) {
        constexpr int mh = decltype(MH)::value, mp = decltype(MP)::value;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const int row = wr * 32 + j * 16 + fr;
                    const int chunk = nh * 32 + wc * 8 + nt * 4 + fq;
                    *(f32x4*)(cl + row * 256 + ((chunk ^ fr) << 2)) = acc[mh][nh][2 * mp + j][nt];
                }
    };

    int tile = blockIdx.x, m0, n0;
    set_tile(tile, m0, n0);
    prologue();
    int iter = 0;
    while (true) {
        STDIFF_STAMP(0);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int d = 0; d < 2; ++d) acc[a][b][c][d] = f32x4{0.f, 0.f, 0.f, 0.f};
        STDIFF_VMCNT(0);
        STDIFF_BARRIER();
        STDIFF_STAMP(1);
        if (wr == 1) STDIFF_BARRIER();                 // stagger the second wave group by one barrier

        for (int t = 0; t < nk; ++t) {
            const int b = t & 1, nb = b ^ 1;
            const bool n1 = t + 1 < nk, n2 = t + 2 < nk;
            // ---- Q1 (A0, B0)
            loadB(b, 0);
            __builtin_amdgcn_sched_barrier(0);
            loadA(b, 0);
            if (n1) dma(nb, 3, t + 1);                 // B1(t+1)
            STDIFF_BARRIER();
            STDIFF_MMA(0, 0);
            STDIFF_BARRIER();
            // ---- Q2 (A0, B1)
            loadB(b, 1);
            if (n1) { dma(nb, 2, t + 1); STDIFF_VMCNT(6); } else { STDIFF_VMCNT(0); }     // B0(t+1); retire A1(t)
            STDIFF_BARRIER();
            STDIFF_MMA(0, 1);
            STDIFF_BARRIER();
            // ---- Q3 (A1, B1)
            loadA(b, 1);
            if (n1) dma(nb, 1, t + 1);                 // A1(t+1)
            STDIFF_BARRIER();
            STDIFF_MMA(1, 1);
            STDIFF_BARRIER();
            // ---- Q4 (A1, B0)
            loadB(b, 0);
            if (n2) { dma(b, 0, t + 2); STDIFF_VMCNT(4); } else if (n1) { STDIFF_VMCNT(2); }   // A0(t+2); retire A0,B1,B0(t+1)
            STDIFF_BARRIER();
            STDIFF_MMA(1, 0);
            STDIFF_BARRIER();
        }
        if (wr == 0) STDIFF_BARRIER();                 // balance the stagger: every wave is past its last LDS read
        STDIFF_STAMP(2);

        const int em0 = m0, en0 = n0, next = tile + (int)gridDim.x;
        if (next < ntiles) {                           // next tile's operands fly in under the epilogue
            set_tile(next, m0, n0);
            prologue();
        }
#pragma unroll 1
        for (int q = 0; q < 4; ++q) {
            if (q) STDIFF_BARRIER();                   // image reads of the previous quarter are done
            switch (q) {
                case 0: put(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}); break;
                case 1: put(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}); break;
                case 2: put(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}); break;
                default: put(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}); break;
            }
            STDIFF_LGKMCNT0();
            STDIFF_BARRIER();
            if (p.vec) {
#define STDIFF_CALL256(E)                                                                  \
    do {                                                                                   \
        if (p.c_bf16) {                                                                    \
            if (p.C2) epi_quarter256<E, true, true>(p, cl, wid, lane, em0, en0, q);        \
            else epi_quarter256<E, true, false>(p, cl, wid, lane, em0, en0, q);            \
        } else {                                                                           \
            if (p.C2) epi_quarter256<E, false, true>(p, cl, wid, lane, em0, en0, q);       \
            else epi_quarter256<E, false, false>(p, cl, wid, lane, em0, en0, q);           \
        }                                                                                  \
    } while (0)
                STDIFF_EPI_DISPATCH(STDIFF_CALL256)
            } else {
                const int rbase = em0 + (q >> 1) * 128 + (q & 1) * 32;
#pragma unroll 1
                for (int ps = 0; ps < 8; ++ps) {
                    const int i = (tid >> 6) + 8 * ps, cq = tid & 63;
                    const int gm = rbase + (i >> 5) * 64 + ((i >> 4) & 1) * 16 + (i & 15), gn = en0 + cq * 4;
                    if (gm < p.M && gn < p.N)
                        epilogue4_slow<bf16_t>(p, gm, gn, *(const f32x4*)(cl + i * 256 + ((cq ^ (i & 15)) << 2)));
                }
            }
            STDIFF_LGKMCNT0();                         // image reads landed before the next put / next tile's DMA
        }
        STDIFF_STAMP(3);
        ++iter;
        if (next >= ntiles) break;
        tile = next;
    }
}

// ---- variant: two 32-MFMA phases per K-slab (see the loop comment); everything else as gemm256p_kernel.
// MODE selects the epilogue code that is compiled in (the kernel sits at 256 VGPRs / 104 SGPRs: every epilogue that is
// merely present costs registers in all of them): 0 = bias / SiLU / GELU / gate*v+res / the generic quarter path,
// 1 = the LayerNorm-modulate epilogue of the adaLN projection only, 2 = gate*v+res that also writes the partial row
// moments of its output.
template <int MODE>
__global__ __launch_bounds__(512, 1) void gemm256q_kernel(Params p) {
    constexpr int ROWB = 128, HALF = 128 * ROWB;          // one half-tile = 128 rows x 64 bf16 = 16 KiB
    __shared__ __attribute__((aligned(16))) unsigned char smem[9 * HALF];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;
    const int fr = lane & 15, fq = lane >> 4;
    const int ntiles = p.tiles_m * p.tiles_n;
    const int nk = p.K / 64;

    // DMA sources: raw-buffer loads straight into LDS.  Half-tile h, piece q = 2*wid + i (8 rows): lane -> row
    // 8q + (lane>>3), 16-byte K-chunk (lane&7) ^ row (the XOR swizzle sits on the source side: the LDS image is
    // lane-linear).  The whole byte offset goes through the VGPR operand so that the descriptor's range check sees
    // it: rows / columns past the edge read as zeros, no clamping.
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(p.A), 0, (int)(((unsigned)(p.M - 1) * p.lda + p.K) * 2u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(p.W), 0, (int)(((unsigned)(p.N - 1) * p.ldw + p.K) * 2u), 0x00020000);
    const int swl = ((lane & 7) ^ (lane >> 3)) << 3;
    const unsigned voffA = (unsigned)((16 * wid + (lane >> 3)) * p.lda + swl) * 2u;
    const unsigned voffW = (unsigned)((16 * wid + (lane >> 3)) * p.ldw + swl) * 2u;
    unsigned sA = 0, sW = 0;       // byte offset of the tile's first row
    bool full = true;              // the unit whose operands are in flight is a whole tile (else: rows m0 .. m0+127)
    auto set_tile = [&](int unit, int& m0, int& n0) {
        int tm, tn, tile = unit, mh = 0;
        full = unit < p.t_full;
        if (!full) { tile = p.t_full + ((unit - p.t_full) >> 1); mh = (unit - p.t_full) & 1; }
        tile_of(tile, ntiles, p.tiles_m, p.tiles_n, tm, tn);
        m0 = tm * 256 + mh * 128; n0 = tn * 256;
        sA = (unsigned)m0 * (unsigned)p.lda * 2u;
        sW = (unsigned)n0 * (unsigned)p.ldw * 2u;
    };
    auto dma = [&](int buf, int slot, int kt) {       // half-tile `slot` of K-slab kt into buffer buf
        unsigned char* dst = smem + (buf * 4 + slot) * HALF + (2 * wid) * 1024;
        const int h = slot & 1;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (slot < 2)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(dst + i * 1024), 16,
                                                         voffA + (sA + (unsigned)((h * 128 + i * 8) * p.lda + kt * 64) * 2u), 0, 0, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(dst + i * 1024), 16,
                                                         voffW + (sW + (unsigned)((h * 128 + i * 8) * p.ldw + kt * 64) * 2u), 0, 0, 0);
        }
    };
    auto prologue = [&]() {                           // slab 0 complete + A0 of slab 1
        dma(0, 0, 0); dma(0, 2, 0); dma(0, 3, 0); dma(0, 1, 0);
        if (nk > 1) dma(1, 0, 1);
    };

    // fragment read offsets inside a half-tile (row & 7 == fr & 7 for every fragment row)
    int offA[2], offB[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int sl = ((ks * 4 + fq) ^ (fr & 7)) << 4;
        offA[ks] = (wr * 64 + fr) * ROWB + sl;
        offB[ks] = (wc * 32 + fr) * ROWB + sl;
    }
    bf16x8 af[4][2], bfr[2][2][2];
    auto loadA = [&](int buf, int mh) {
        const unsigned char* b = smem + (buf * 4 + mh) * HALF;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) af[mt][ks] = *(const bf16x8*)(b + offA[ks] + mt * 16 * ROWB);
    };
    auto loadB = [&](int buf, int nh) {
        const unsigned char* b = smem + (buf * 4 + 2 + nh) * HALF;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) bfr[nh][nt][ks] = *(const bf16x8*)(b + offB[ks] + nt * 16 * ROWB);
    };

    f32x4 acc[2][2][4][2];

#define STDIFF_MMA(MH, NH)                                                                                    \
    do {                                                                                                      \
        __builtin_amdgcn_s_setprio(1);                                                                        \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                      \
        _Pragma("unroll") for (int mt = 0; mt < 4; ++mt)                                                      \
        _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                                      \
            acc[MH][NH][mt][nt] =                                                                             \
                STDIFF_MFMA_16x16x32_H16(bfr[NH][nt][ks], af[mt][ks], acc[MH][NH][mt][nt], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                                        \
    } while (0)

    // C quarter image (slots 5..8), XOR-swizzled 16-byte chunks
    float* cl = (float*)(smem + 5 * HALF);
    const unsigned cla = lds_addr(cl);
    auto put = [&](auto MH, auto MP) {
        constexpr int mh = decltype(MH)::value, mp = decltype(MP)::value;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int nh = 0; nh < 2; ++nh)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const int row = wr * 32 + j * 16 + fr;
                    const int chunk = nh * 32 + wc * 8 + nt * 4 + fq;
                    *(f32x4*)(cl + row * 256 + ((chunk ^ fr) << 2)) = acc[mh][nh][2 * mp + j][nt];
                }
    };

    int tile = blockIdx.x, m0, n0;
    set_tile(tile, m0, n0);
    prologue();
    int iter = 0;
    while (true) {
        STDIFF_STAMP(0);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int d = 0; d < 2; ++d) acc[a][b][c][d] = f32x4{0.f, 0.f, 0.f, 0.f};
        STDIFF_BARRIER();                              // every wave is done with the C image (slots 5..8)
        if (nk > 1) { dma(1, 2, 1); dma(1, 3, 1); STDIFF_VMCNT(4); } else { STDIFF_VMCNT(0); }   // B0,B1 of slab 1
        STDIFF_BARRIER();
        STDIFF_STAMP(1);
        if (wr == 1) STDIFF_BARRIER();                 // stagger the second wave group by one barrier

        // Two phases per K-slab, 32 MFMAs each: P1 = A0 x (B0, B1), P2 = A1 x (B0, B1) with the B fragments of both
        // column halves kept in registers (4 barriers per slab instead of 8, 24 LDS reads instead of 28).  Refills two
        // slabs ahead: {A0,B0,B1}(t+2) in P2 of slab t (every wave has read them by then), A1(t+2) in P1 of slab t+1.
        // Both counted waits are vmcnt(8) in steady state: they leave [A1 | A0 B0 B1 | A1] resp. [A1 | A0 B0 B1] in flight.
        for (int t = 0; t < nk; ++t) {
            const int b = t & 1, nb = b ^ 1;
            const bool n1 = t + 1 < nk, n2 = t + 2 < nk;
            // ---- P1: A0 x (B0, B1)
            loadB(b, 0);
            loadB(b, 1);
            __builtin_amdgcn_sched_barrier(0);
            loadA(b, 0);
            if (n1) { dma(nb, 1, t + 1); STDIFF_VMCNT(8); } else { STDIFF_VMCNT(0); }      // A1(t+1); retire A1(t)
            STDIFF_BARRIER();
            STDIFF_MMA(0, 0);
            STDIFF_MMA(0, 1);
            STDIFF_BARRIER();
            // ---- P2: A1 x (B0, B1); a half unit keeps the DMA schedule (its A1 rows are loaded and ignored)
            if (full) loadA(b, 1);
            if (n2) { dma(b, 0, t + 2); dma(b, 2, t + 2); dma(b, 3, t + 2); STDIFF_VMCNT(8); }   // retire A0,B0,B1(t+1)
            else if (n1) { STDIFF_VMCNT(2); }
            STDIFF_BARRIER();
            if (full) {
                STDIFF_MMA(1, 0);
                STDIFF_MMA(1, 1);
            }
            STDIFF_BARRIER();
        }
        if (wr == 0) STDIFF_BARRIER();                 // balance the stagger: every wave is past its last LDS read
        STDIFF_STAMP(2);

        const int em0 = m0, en0 = n0, next = tile + (int)gridDim.x;
        const int nq = full ? 4 : 2;                   // quarters of this unit (a half unit holds rows in acc[0] only)
        if (next < p.units) {                           // next tile's operands fly in under the epilogue
            set_tile(next, m0, n0);
            prologue();
        }
#ifdef STDIFF_GEMM_NOEPI      // timing experiment only: the tile's results are dropped (upper bound of any epilogue overlap)
        _Pragma("unroll") for (int a0 = 0; a0 < 2; ++a0) _Pragma("unroll") for (int a1 = 0; a1 < 2; ++a1)
        _Pragma("unroll") for (int a2 = 0; a2 < 4; ++a2) _Pragma("unroll") for (int a3 = 0; a3 < 2; ++a3)
            asm volatile("" ::"v"(acc[a0][a1][a2][a3]));
        if (p.M >= 0) {} else
#endif
        if constexpr (MODE == 1) {
            epi_tile256_lnmod(p, acc, cla, tid, wid, lane, em0, en0, nq, iter);
        } else if constexpr (MODE == 2) {
            epi_tile256_gate_res<false, true, 3>(p, acc, cla, wid, lane, em0, en0, nq);
        } else if (p.vec == 2 && p.c_bf16 && !p.C2 && p.epi <= STDIFF_EPI_GELU) {     // bias / SiLU / GELU, 16-byte C rows
            unsigned char* img = smem + 5 * HALF;
            if (p.epi == STDIFF_EPI_GELU) epi_tile256_bf16<STDIFF_EPI_GELU>(p, acc, img, tid, wid, lane, em0, en0, nq);
            else if (p.epi == STDIFF_EPI_SILU) epi_tile256_bf16<STDIFF_EPI_SILU>(p, acc, img, tid, wid, lane, em0, en0, nq);
            else epi_tile256_bf16<STDIFF_EPI_BIAS>(p, acc, img, tid, wid, lane, em0, en0, nq);
        } else if (p.vec && p.epi == STDIFF_EPI_GATE_RES && !p.C2) {
            if (p.c_bf16) epi_tile256_gate_res<true>(p, acc, cla, wid, lane, em0, en0, nq);
            else epi_tile256_gate_res<false>(p, acc, cla, wid, lane, em0, en0, nq);
        } else
#pragma unroll 1
        for (int q = 0; q < nq; ++q) {
            if (q) STDIFF_BARRIER();                   // image reads of the previous quarter are done
            switch (q) {
                case 0: put(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}); break;
                case 1: put(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}); break;
                case 2: put(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}); break;
                default: put(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}); break;
            }
            STDIFF_LGKMCNT0();
            STDIFF_BARRIER();
            if (p.vec) {
#define STDIFF_CALL256(E)                                                                  \
    do {                                                                                   \
        if (p.c_bf16) {                                                                    \
            if (p.C2) epi_quarter256<E, true, true>(p, cl, wid, lane, em0, en0, q);        \
            else epi_quarter256<E, true, false>(p, cl, wid, lane, em0, en0, q);            \
        } else {                                                                           \
            if (p.C2) epi_quarter256<E, false, true>(p, cl, wid, lane, em0, en0, q);       \
            else epi_quarter256<E, false, false>(p, cl, wid, lane, em0, en0, q);           \
        }                                                                                  \
    } while (0)
                STDIFF_EPI_DISPATCH(STDIFF_CALL256)
            } else {
                const int rbase = em0 + (q >> 1) * 128 + (q & 1) * 32;
#pragma unroll 1
                for (int ps = 0; ps < 8; ++ps) {
                    const int i = (tid >> 6) + 8 * ps, cq = tid & 63;
                    const int gm = rbase + (i >> 5) * 64 + ((i >> 4) & 1) * 16 + (i & 15), gn = en0 + cq * 4;
                    if (gm < p.M && gn < p.N)
                        epilogue4_slow<bf16_t>(p, gm, gn, *(const f32x4*)(cl + i * 256 + ((cq ^ (i & 15)) << 2)));
                }
            }
            STDIFF_LGKMCNT0();                         // image reads landed before the next put / next tile's DMA
        }
        STDIFF_STAMP(3);
        ++iter;
        if (next >= p.units) break;
        tile = next;
    }
}

// ================================================================ 16-bit, 128 x 128 tile, LDS-DMA staging (batch-sized GEMMs)
// gemm_nt_kernel stages its operands through registers (global -> VGPR -> ds_write): on a grid of one workgroup per CU
// that path is what bounds the K loop (0.69 us per 64-wide slab against 0.21 us of MFMA: profiles/notes/
// r03_training_step.md).  This variant of the same tile -- same fragment reads, same MFMA order, the same bits -- fills
// a ring of four slab buffers by LDS-DMA instead (global_load_lds, 16 B per lane, no VGPR round trip, no ds_write),
// three slabs ahead, with ONE barrier per slab and counted waits.  The LDS image is lane-linear, so the XOR swizzles of
// the register-staged kernel are applied to the per-lane SOURCE address.  Row-major A; W row-major or K-major (WK: the
// input-gradient GEMMs); rows / columns past the edges and K-major rows k >= k_valid read as zeros through the buffer
// descriptor's range check (columns past N of a K-major W: an offset forced out of range).  One workgroup per CU
// (128 KiB of LDS): selected for the grids gemm_nt_kernel runs with three slabs of loads in flight.
// NW = 8: eight waves (2 x 4, a wave owns 64 x 32 outputs) instead of four (2 x 2, 64 x 64): two waves per SIMD, so one
// wave's fragment reads run under the other's MFMAs (with one wave per SIMD they alternate: 0.25 us of LDS reads +
// 0.25 us of MFMA per slab).  Every output element still sums its products in the same order: same bits.
template <bool WK, int NW>
__global__ __launch_bounds__(512, 1) void gemm128d_kernel(Params p) {
    using T = bf16_t;
    constexpr int TILEB = 16384, STAGES = 4;       // one operand tile: 128 rows x 128 B, or 64 k-rows x 256 B
    constexpr int NTW = NW == 8 ? 2 : 4;           // 16-column fragments per wave; DMA instructions per wave and operand: 16 / NW
    constexpr int DPW = 16 / NW;
    __shared__ __attribute__((aligned(16))) unsigned char smem[STAGES * 2 * TILEB];
    static_assert(STAGES * 2 * TILEB >= BM * CLD * 4, "the fp32 C image reuses the ring");

    int tm, tn;
    const int ntiles = p.tiles_m * p.tiles_n;
    const int bid = blockIdx.x;
    const int split = bid / ntiles;
    tile_of(bid - split * ntiles, ntiles, p.tiles_m, p.tiles_n, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = NW == 8 ? wid >> 2 : wid >> 1, wn = NW == 8 ? wid & 3 : wid & 1;
    const int wcol = wn * (NW == 8 ? 32 : 64);
    const int fr = lane & 15, fq = lane >> 4;

    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(p.A), 0, (int)(((unsigned)(p.M - 1) * p.lda + p.K) * 2u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(p.W), 0,
        WK ? (int)(((unsigned)(p.k_valid - 1) * p.ldw + p.N) * 2u) : (int)(((unsigned)(p.N - 1) * p.ldw + p.K) * 2u), 0x00020000);
    // per-lane source offsets (bytes) of this wave's DMA instructions per operand and slab; instruction j fills bytes
    // [(DPW wid + j) KiB, + 1 KiB) of the tile image
    unsigned voffA[4], voffW[4];           // (DPW of them used)
#pragma unroll
    for (int j = 0; j < DPW; ++j) {
        const int row = 8 * (DPW * wid + j) + (lane >> 3), slot = (lane & 7) ^ (lane >> 3);      // row & 7 == lane >> 3
        voffA[j] = ((unsigned)(m0 + row) * (unsigned)p.lda + slot * 8) * 2u;
        if constexpr (WK) {
            const int k = 4 * (DPW * wid + j) + (lane >> 4), pp = lane & 15;
            const int fk = (k & 3) | (((k >> 3) & 1) << 2);
            const int piece = ((((pp >> 1) ^ fk)) << 1) | (pp & 1);
            const int n = n0 + piece * 8;
            voffW[j] = n < p.N ? ((unsigned)k * (unsigned)p.ldw + n) * 2u : 0x80000000u;
        } else {
            voffW[j] = ((unsigned)(n0 + row) * (unsigned)p.ldw + slot * 8) * 2u;
        }
    }
    const int kb = split * p.nk_per, nk = min(p.K / 64 - kb, p.nk_per);     // this block's K-slabs
    auto dma = [&](int slab, int stage) {
        unsigned char* da = smem + stage * 2 * TILEB + (DPW * wid) * 1024;
        unsigned char* dw = da + TILEB;
        const unsigned ka = (unsigned)(kb + slab) * 128u;                                  // 64 k x 2 B along a row
        const unsigned kw = WK ? (unsigned)(kb + slab) * 64u * (unsigned)p.ldw * 2u : ka;  // 64 k-rows
#pragma unroll
        for (int j = 0; j < DPW; ++j) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(da + j * 1024), 16,
                                                     voffA[j] + ka, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (__attribute__((address_space(3))) void*)(dw + j * 1024), 16,
                                                     (WK && voffW[j] == 0x80000000u) ? 0x80000000u : voffW[j] + kw, 0, 0, 0);
        }
    };

    f32x4 acc[4][4];              // (NW = 8 uses acc[.][0..1])
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragments exactly as gemm_nt_body reads them (row-major: 16-byte reads; K-major: transposed 4 x 16 block reads)
    auto frags = [&](int stage, int ks, bf16x8 (&af)[4], bf16x8 (&wf)[4]) {
        const unsigned char* ba = smem + stage * 2 * TILEB;
        const unsigned char* bw = ba + TILEB;
        const int slot = ks * 4 + fq;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int ra_ = wm * 64 + t * 16 + fr;
            af[t] = *(const bf16x8*)(ba + ra_ * 128 + ((slot ^ (ra_ & 7)) << 4));
        }
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            const int rw_ = wcol + t * 16 + fr;
            if constexpr (WK) {
                const int kk = ks * 32 + 8 * fq + (fr >> 2), fk = (kk & 3) | (((kk >> 3) & 1) << 2);
                const unsigned char* w0 = bw + kk * 256 + ((((wcol + t * 16) >> 4) ^ fk) << 5) + 8 * (fr & 3);
                bf16x4 lo = ds_read_tr16_b64(w0);
                bf16x4 hi = ds_read_tr16_b64(w0 + 4 * 256);
                wf[t] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            } else {
                wf[t] = *(const bf16x8*)(bw + rw_ * 128 + ((slot ^ (rw_ & 7)) << 4));
            }
        }
    };
    auto mma = [&](const bf16x8 (&af)[4], const bf16x8 (&wf)[4]) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt)
                acc[mt][nt] = STDIFF_MFMA_16x16x32_H16(wf[nt], af[mt], acc[mt][nt], 0, 0, 0);
    };

    // Ring of four: slabs k + 1, k + 2 (and k + 3, issued behind the barrier) are in flight while slab k is multiplied; ONE
    // barrier per slab.  (Pipelining the fragment reads across a barrier in the middle of the slab, as gemm_nt_body's
    // deep variant does, was tried here and lost: 15.9 against 13.7 us for a K = 1024 launch.)
#pragma unroll
    for (int d = 0; d < STAGES - 1; ++d)
        if (d < nk) dma(d, d);
#pragma unroll 1
    for (int k = 0; k < nk; ++k) {
        const int rem = min(STAGES - 2, nk - 1 - k);       // slabs issued after slab k that may still be in flight
        if constexpr (NW == 8) {
            if (rem >= 2) STDIFF_VMCNT(8);
            else if (rem == 1) STDIFF_VMCNT(4);
            else STDIFF_VMCNT(0);
        } else {
            if (rem >= 2) STDIFF_VMCNT(16);
            else if (rem == 1) STDIFF_VMCNT(8);
            else STDIFF_VMCNT(0);
        }
        STDIFF_BARRIER();                                  // slab k is complete; every wave is done with slab k - 1
        if (k + STAGES - 1 < nk) dma(k + STAGES - 1, (k + STAGES - 1) & (STAGES - 1));
        const int stage = k & (STAGES - 1);
        bf16x8 af0[4], wf0[4], af1[4], wf1[4];
        frags(stage, 0, af0, wf0);
        frags(stage, 1, af1, wf1);
        __builtin_amdgcn_sched_barrier(0);
        mma(af0, wf0);
        mma(af1, wf1);
    }
    STDIFF_BARRIER();                                      // the last slab has been read: the C image may overwrite the ring

    // ---- accumulators -> fp32 C image in LDS and out, as in gemm_nt_body (the first 256 threads run its epilogue)
    float* cl = (float*)smem;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < NTW; ++nt) {
            const int m = wm * 64 + mt * 16 + fr, n = wcol + nt * 16 + fq * 4;
            *(f32x4*)(cl + m * CLD + n) = acc[mt][nt];
        }
    __syncthreads();
    if (NW == 8 && tid >= NT) return;
    if (p.ksplit > 1) {                 // partial product -> workspace slice (splitk_reduce_kernel or a *_split consumer)
        float* w = p.ws + (size_t)split * p.M * p.ldws;
#pragma unroll 1
        for (int i = 0; i < 16; ++i) {
            const int r = (tid >> 5) + 8 * i, c4 = (tid & 31) * 4;
            const int gm = m0 + r, gn = n0 + c4;
            if (gm < p.M && gn < p.ldws) *(f32x4*)(w + (size_t)gm * p.ldws + gn) = *(const f32x4*)(cl + r * CLD + c4);
        }
        return;
    }
    if (p.vec) {
#define STDIFF_CALL128D(E) epi_rows128<E, T>(p, cl, tid, m0, n0)
        STDIFF_EPI_DISPATCH(STDIFF_CALL128D)
        return;
    }
#pragma unroll 1
    for (int i = 0; i < 16; ++i) {
        const int r = (tid >> 4) + 16 * (i >> 1), c4 = (tid & 15) * 8 + 4 * (i & 1);
        const int gm = m0 + r, gn = n0 + c4;
        if (gm < p.M && gn < p.N) epilogue4_slow<T>(p, gm, gn, *(const f32x4*)(cl + r * CLD + c4));
    }
}

// ================================================================ 16-bit, 192 x 256 tile, TWO tiles in flight per CU
// gemm256q spends 17-29 % of a tile in its epilogue (LDS round trips, barriers, store issue) with the matrix cores idle,
// and cannot overlap it with the next tile: one 256 x 256 fp32 tile is 128 of a wave's 256 registers.  This kernel makes
// the tile 192 x 256 (96 accumulator registers per wave) so that the PREVIOUS tile fits next to the current one:
//   * at a tile's last MFMA each wave applies bias / SiLU / GELU to its accumulators and packs the result to the 16-bit
//     format (48 registers) -- the only epilogue work that is still exposed (VALU only, no memory, no barrier);
//   * the packed tile leaves during the NEXT tile's K loop, in 6 steps of 64 rows x 128 columns: every wave puts its
//     piece into a 16 KiB LDS image in the pre-barrier part of one K-slab phase, and one slab later the four
//     "storer" waves copy the image to C as 256-byte row pieces (16-byte stores, whole 128-byte lines);
//   * the operand ring (LDS-DMA, two K-slabs ahead, counted vmcnt waits) runs on ACROSS tile boundaries, so a tile starts
//     with its first two slabs already in LDS.
// Wave roles.  8 waves = 4 (M: 48 rows) x 2 (N: 64 columns of each 128-column half).  Waves 0-3 (wn = 0) issue every
// LDS-DMA load and are the only ones that wait on vmcnt; waves 4-7 (wn = 1) issue every global store and never wait on
// vmcnt: loads complete in order among loads, but stores may complete out of order with them, so a wave that had both
// in flight could not use counted waits.  Waves 4-7 run one barrier behind waves 0-3 (as in gemm256q), so on every SIMD
// one wave is in its MFMA cluster while its partner reads fragments.
// Per K-slab (64): P1 = A x B0, P2 = A x B1, 24 MFMAs (16x16x32) each per wave; the A fragments of the slab stay in
// registers for both phases.  Refills: P1(t) issues B1(t+1), P2(t) issues {A, B0}(t+2); both waits are vmcnt(14).
// Same accumulation order and epilogue arithmetic as the other 16-bit kernels: the same bits for the same row.
// LDS: 2 x (A 24 KiB + B0 16 KiB + B1 16 KiB) + 2 x 16 KiB image + 1 KiB bias = 145 KiB.
__device__ __forceinline__ bf16x8 lds_read_frag(unsigned addr) {
    bf16x8 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}

#ifdef STDIFF_T192_STAMPS
// debug build only (scripts/gemm192_stamps.py): shader-clock stamps of waves 0 and 4 of workgroup 0 over its first 64
// K-slabs, 10 points per slab: 0 slab start, 1 P1 reads + DMA issued, 2 after the vmcnt / lgkmcnt waits, 3 after barrier,
// 4 after the MFMAs, 5 after barrier, 6 P2 reads + DMA issued, 7 after waits + barrier, 8 after MFMAs, 9 after barrier
__device__ unsigned long long g_t192[2 * 64 * 10];
#define STDIFF_TS(k)                                                                                          \
    do {                                                                                                      \
        if (blockIdx.x == 0 && lane == 0 && (wid & 3) == 0 && gslab < 64)                                     \
            g_t192[((wid >> 2) * 64 + gslab) * 10 + (k)] = __builtin_amdgcn_s_memtime();                      \
    } while (0)
#else
#define STDIFF_TS(k) do {} while (0)
#endif

template <int EPI>
__global__ __launch_bounds__(512, 1) void gemm192t_kernel(Params p) {
    constexpr int ROWB = 128;
    constexpr int A_BYTES = 192 * ROWB, B_BYTES = 128 * ROWB, BUF = A_BYTES + 2 * B_BYTES;
    constexpr int IMG_OFF = 2 * BUF, IMG_BYTES = 64 * 256, BIAS_OFF = IMG_OFF + 2 * IMG_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char smem[BIAS_OFF + 1024];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid & 3, wn = wid >> 2;
    const bool issuer = wn == 0;
    const int fr = lane & 15, fq = lane >> 4;
    const int ntiles = p.tiles_m * p.tiles_n;
    const int nk = p.K / 64;
    const unsigned lds0 = lds_addr(smem);

    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(p.A), 0, (int)(((unsigned)(p.M - 1) * p.lda + p.K) * 2u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(p.W), 0, (int)(((unsigned)(p.N - 1) * p.ldw + p.K) * 2u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.bias), 0, p.bias ? p.N * 4 : 0, 0x00020000);      // no bias: every read is out of range = 0
    // DMA lane part: an instruction moves 8 rows x 128 B; lane -> row lane>>3, 16-byte K-chunk (lane&7) ^ row (the XOR
    // swizzle sits on the source side: the LDS image is lane-linear)
    const int swl = ((lane & 7) ^ (lane >> 3)) << 3;
    const unsigned vA = (unsigned)((lane >> 3) * p.lda + swl) * 2u, vW = (unsigned)((lane >> 3) * p.ldw + swl) * 2u;
    const unsigned stepA = (unsigned)p.lda * 16u, stepW = (unsigned)p.ldw * 16u;      // bytes per 8-row group

    // issuer wave w: A groups 6w .. 6w+5 (of 24), B half groups 4w .. 4w+3 (of 16)
    auto dmaA = [&](int buf, unsigned base) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int g = wm * 6 + i;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (__attribute__((address_space(3))) void*)(smem + buf * BUF + g * 1024), 16,
                                                     vA + base + (unsigned)g * stepA, 0, 0, 0);
        }
    };
    auto dmaB = [&](int buf, int nh, unsigned base) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int g = wm * 4 + i;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(
                rsW, (__attribute__((address_space(3))) void*)(smem + buf * BUF + A_BYTES + nh * B_BYTES + g * 1024), 16,
                vW + base + (unsigned)(nh * 16 + g) * stepW, 0, 0, 0);
        }
    };

    // fragment addresses (row & 7 == fr & 7 for every fragment row: 48 wm, 64 wn and 16 mt / nt are multiples of 8... 48 is)
    unsigned offA[2], offB[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int sl = ((ks * 4 + fq) ^ (fr & 7)) << 4;
        offA[ks] = lds0 + (unsigned)((wm * 48 + fr) * ROWB + sl);
        offB[ks] = lds0 + (unsigned)(A_BYTES + (wn * 64 + fr) * ROWB + sl);
    }
    bf16x8 af[3][2], bfr[4][2];
    f32x4 acc[2][3][4];
    u32x2 prev[2][3][4];

#define STDIFF_T_LOADA(bufoff)                                                                          \
    _Pragma("unroll") for (int mt = 0; mt < 3; ++mt) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)      \
        af[mt][ks] = lds_read_frag(offA[ks] + (bufoff) + mt * 16 * ROWB)
#define STDIFF_T_LOADB(bufoff, NH)                                                                      \
    _Pragma("unroll") for (int nt = 0; nt < 4; ++nt) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)      \
        bfr[nt][ks] = lds_read_frag(offB[ks] + (bufoff) + (NH) * B_BYTES + nt * 16 * ROWB)
    // every fragment register passes through the wait, so no MFMA can be scheduled above it
#define STDIFF_T_WAIT_A()                                                                                    \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                                      \
                 : "+v"(af[0][0]), "+v"(af[0][1]), "+v"(af[1][0]), "+v"(af[1][1]), "+v"(af[2][0]), "+v"(af[2][1])::"memory")
#define STDIFF_T_WAIT_B()                                                                                    \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                                      \
                 : "+v"(bfr[0][0]), "+v"(bfr[0][1]), "+v"(bfr[1][0]), "+v"(bfr[1][1]), "+v"(bfr[2][0]),      \
                   "+v"(bfr[2][1]), "+v"(bfr[3][0]), "+v"(bfr[3][1])::"memory")
#define STDIFF_T_MMA(NH)                                                                                \
    do {                                                                                                \
        __builtin_amdgcn_s_setprio(1);                                                                  \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                \
        _Pragma("unroll") for (int mt = 0; mt < 3; ++mt)                                                \
        _Pragma("unroll") for (int nt = 0; nt < 4; ++nt)                                                \
            acc[NH][mt][nt] = STDIFF_MFMA_16x16x32_H16(bfr[nt][ks], af[mt][ks], acc[NH][mt][nt], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                                  \
    } while (0)

    // ---- the previous tile on its way out
    int pm0 = 0, pn0 = 0;
    // step s = 2 mt + nh: every wave puts rows {16 wm + fr} x columns {64 wn + 16 nt + 4 fq ..+3} of (mt, nh) into image
    // s & 1.  Image row r (256 B) holds 32 8-byte chunks; chunk c sits at c ^ ((r & 15) << 1): the 32 lanes of a
    // half-wave write 32 distinct bank pairs, and 16-byte read units stay adjacent (bit 0 is never flipped)
    auto img_write = [&](auto S) {
        constexpr int s = decltype(S)::value, mt = s >> 1, nh = s & 1;
        int l = lane;
        asm volatile("" : "+v"(l));          // addresses are recomputed at every step, not kept live across the K loop
        const int fr_ = l & 15, fq_ = l >> 4;
        const unsigned base = lds0 + IMG_OFF + (s & 1) * IMG_BYTES + (unsigned)((wm * 16 + fr_) * 256);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            const int c = wn * 16 + nt * 4 + fq_;
            lds_write8(base + (unsigned)((c ^ (fr_ << 1)) << 3), prev[nh][mt][nt]);
        }
    };
    // storer wave k = wid - 4 copies image rows 16 k .. 16 k + 15 (tile rows 48 k + 16 mt + 0..15), 4 rows per instruction
    auto img_store = [&](auto S) {
        constexpr int s = decltype(S)::value, mt = s >> 1, nh = s & 1;
        int l = lane;
        asm volatile("" : "+v"(l));
        const int u = l & 15, gn = pn0 + nh * 128 + u * 8;
        u32x4 v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = wm * 16 + i * 4 + (l >> 4);
            const f32x4 t = lds_read16(lds0 + IMG_OFF + (s & 1) * IMG_BYTES + (unsigned)(row * 256 + (((2 * u) ^ ((row & 15) << 1)) << 3)));
            v[i] = __builtin_bit_cast(u32x4, t);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3])::"memory");
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int gm = pm0 + wm * 48 + mt * 16 + i * 4 + (l >> 4);
            if (gm < p.M && gn < p.N)
                __builtin_nontemporal_store(v[i], (u32x4*)((bf16_t*)p.C + (size_t)gm * p.ldc + gn));
        }
    };
    // bias / SiLU / GELU on the accumulators, packed to the 16-bit format: the same arithmetic as epi_tile256_bf16
    auto pack = [&]() {
        f32x4 b4[2][4];
        int l = lane;
        asm volatile("" : "+v"(l));
#pragma unroll
        for (int nh = 0; nh < 2; ++nh)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
                b4[nh][nt] = lds_read16(lds0 + BIAS_OFF + (unsigned)((nh * 128 + wn * 64 + nt * 16 + (l >> 4) * 4) * 4));
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(b4[0][0]), "+v"(b4[0][1]), "+v"(b4[0][2]), "+v"(b4[0][3]), "+v"(b4[1][0]), "+v"(b4[1][1]),
                       "+v"(b4[1][2]), "+v"(b4[1][3])::"memory");
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int nh = 0; nh < 2; ++nh)
#pragma unroll
            for (int mt = 0; mt < 3; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) {
                    const f32x4 o = epi_math<EPI, true>(acc[nh][mt][nt] + b4[nh][nt], zero, zero);
                    u32x2 rr;
                    rr[0] = pack2bf(o[0], o[1]);
                    rr[1] = pack2bf(o[2], o[3]);
                    prev[nh][mt][nt] = rr;
                    acc[nh][mt][nt] = zero;
                }
    };
    auto dma_bias = [&](int n0) {          // 256 floats = one instruction; lanes past N read zeros
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)(smem + BIAS_OFF), 16,
                                                 (unsigned)(n0 + lane * 4) * 4u, 0, 0, 0);
    };

    // ---- tile walk
    int tile = blockIdx.x, m0, n0, nm0 = 0, nn0 = 0;
    {
        int tm, tn;
        tile_of(tile, ntiles, p.tiles_m, p.tiles_n, tm, tn);
        m0 = tm * 192; n0 = tn * 256;
    }
    int next = tile + (int)gridDim.x;
    bool has_next = next < ntiles, has_prev = false;
    if (has_next) {
        int tm, tn;
        tile_of(next, ntiles, p.tiles_m, p.tiles_n, tm, tn);
        nm0 = tm * 192; nn0 = tn * 256;
    }
    // byte offset of K-slab `kt` of the tile in flight (kt >= nk: the next tile's slab kt - nk)
    auto baseA = [&](int kt) { return kt < nk ? (unsigned)m0 * (unsigned)p.lda * 2u + (unsigned)kt * 128u
                                              : (unsigned)nm0 * (unsigned)p.lda * 2u + (unsigned)(kt - nk) * 128u; };
    auto baseW = [&](int kt) { return kt < nk ? (unsigned)n0 * (unsigned)p.ldw * 2u + (unsigned)kt * 128u
                                              : (unsigned)nn0 * (unsigned)p.ldw * 2u + (unsigned)(kt - nk) * 128u; };

#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[a][b][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (issuer) {                       // slab 0 complete, {A, B0} of slab 1 (nk >= 13 is a launch condition)
        dmaA(0, baseA(0)); dmaB(0, 0, baseW(0)); dmaB(0, 1, baseW(0));
        dmaA(1, baseA(1)); dmaB(1, 0, baseW(1));
        if (wid == 0) dma_bias(n0);
        STDIFF_VMCNT(10);               // wave 0 has one more in flight (the bias): its wait is only stronger
    }
    STDIFF_BARRIER();
    if (wn == 1) STDIFF_BARRIER();      // stagger the second wave group by one barrier

    int gslab = 0;
    while (true) {
        for (int t = 0; t < nk; ++t, ++gslab) {
            const unsigned boff = (unsigned)((t & 1) * BUF), nboff = (unsigned)(((t + 1) & 1) * BUF);
            const bool more1 = t + 1 < nk || has_next, more2 = t + 2 < nk || has_next;
            STDIFF_TS(0);
            // ---- P1: A x B0
            STDIFF_T_LOADA(boff);
            STDIFF_T_LOADB(boff, 0);
            if (has_prev) {
                switch (t) {
                    case 1: img_write(std::integral_constant<int, 0>{}); break;
                    case 3: img_write(std::integral_constant<int, 1>{}); break;
                    case 5: img_write(std::integral_constant<int, 2>{}); break;
                    case 7: img_write(std::integral_constant<int, 3>{}); break;
                    case 9: img_write(std::integral_constant<int, 4>{}); break;
                    case 11: img_write(std::integral_constant<int, 5>{}); break;
                    default: break;
                }
                if (!issuer) {
                    switch (t) {
                        case 2: img_store(std::integral_constant<int, 0>{}); break;
                        case 4: img_store(std::integral_constant<int, 1>{}); break;
                        case 6: img_store(std::integral_constant<int, 2>{}); break;
                        case 8: img_store(std::integral_constant<int, 3>{}); break;
                        case 10: img_store(std::integral_constant<int, 4>{}); break;
                        case 12: img_store(std::integral_constant<int, 5>{}); break;
                        default: break;
                    }
                }
            }
            if (issuer) {
                if (more1) { dmaB((t + 1) & 1, 1, baseW(t + 1)); STDIFF_TS(1); STDIFF_VMCNT(14); }      // B1(t+1); retires B1(t)
                else { STDIFF_VMCNT(0); }
            } else { STDIFF_TS(1); }
            STDIFF_T_WAIT_A();
            STDIFF_T_WAIT_B();
            STDIFF_TS(2);
            STDIFF_BARRIER();
            STDIFF_TS(3);
            STDIFF_T_MMA(0);
            STDIFF_TS(4);
            STDIFF_BARRIER();
            STDIFF_TS(5);
            // ---- P2: A x B1
            STDIFF_T_LOADB(boff, 1);
            if (issuer) {
                // this tile's bias goes out HERE, not at the tile switch: the second wave group packs the previous tile
                // (reading its bias) one barrier after wave 0 does, and has passed that point by now
                if (t == 0 && has_prev && wid == 0) dma_bias(n0);
                if (more2) { dmaA(t & 1, baseA(t + 2)); dmaB(t & 1, 0, baseW(t + 2)); STDIFF_VMCNT(14); }   // retires {A,B0}(t+1)
                else if (more1) { STDIFF_VMCNT(4); }
            }
            STDIFF_TS(6);
            STDIFF_T_WAIT_B();
            STDIFF_BARRIER();
            STDIFF_TS(7);
            STDIFF_T_MMA(1);
            STDIFF_TS(8);
            STDIFF_BARRIER();
            STDIFF_TS(9);
            (void)nboff;
        }
        pack();
        pm0 = m0; pn0 = n0; has_prev = true;
        if (!has_next) break;
        tile = next; m0 = nm0; n0 = nn0;
        next = tile + (int)gridDim.x;
        has_next = next < ntiles;
        if (has_next) {
            int tm, tn;
            tile_of(next, ntiles, p.tiles_m, p.tiles_n, tm, tn);
            nm0 = tm * 192; nn0 = tn * 256;
        }
    }
    if (wn == 0) STDIFF_BARRIER();       // balance the stagger
    // ---- the last tile leaves serially
#define STDIFF_T_DRAIN(S)                                           \
    do {                                                            \
        img_write(std::integral_constant<int, S>{});                \
        STDIFF_LGKMCNT0();                                          \
        STDIFF_BARRIER();                                           \
        if (!issuer) img_store(std::integral_constant<int, S>{});   \
    } while (0)
    STDIFF_T_DRAIN(0); STDIFF_T_DRAIN(1); STDIFF_T_DRAIN(2); STDIFF_T_DRAIN(3); STDIFF_T_DRAIN(4); STDIFF_T_DRAIN(5);
}

// ---------------------------------------------------------------- transpose (bf16 / f32), 64x64 tiles through LDS
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const T* in, int ldi, T* out, int ldo, int rows, int cols,
                                                        int rows_pad) {
    __shared__ T tile[64][65];
    int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < rows && c < cols) ? in[(size_t)r * ldi + c] : (T)0;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        int c = c0 + i, r = r0 + tx;
        if (r < rows_pad && c < cols) out[(size_t)c * ldo + r] = tile[tx][i];   // rows..rows_pad-1: zeros
    }
}

inline bool al16(const void* p) { return ((uintptr_t)p & 15) == 0; }
inline bool al8(const void* p) { return ((uintptr_t)p & 7) == 0; }

}  // namespace

// Split-K scratch that is part of the LIBRARY IMAGE instead of a caller allocation (stdiff_gemm_static_scratch): one
// copy per GPU the code object is loaded on, held until the process ends.  In host code the array's name is only a
// stand-in for the symbol; its device address comes from hipGetSymbolAddress, per device.
static __device__ __align__(256) float scratch[STDIFF_STATIC_SCRATCH_BYTES / sizeof(float)];

namespace {
int gemm_impl(const stdiff_gemm_args* a, void* stream, void* workspace, long long workspace_bytes, bool strict_ws = false);

// The STDIFF_GEMM_* / STDIFF_SPLITK_* knobs (A/B timing and tests), read from the environment ONCE per process -- not
// per launch -- into an immutable record; stdiff_reload_env() publishes a fresh record (tests that switch kernels in
// one process).  Readers never see a half-written record: the pointer is swapped atomically and old records are left
// alive (a few dozen bytes each).
struct GemmEnv {
    int tile, cus, split_tail, splitk_max, splitk_blocks, splitk_min, depth, dma128;
};
const GemmEnv* read_gemm_env() {
    auto num = [](const char* name, int dflt) { const char* e = getenv(name); return e && *e ? atoi(e) : dflt; };
    GemmEnv* e = new GemmEnv;
    e->tile = num("STDIFF_GEMM_TILE", 0);
    e->cus = num("STDIFF_GEMM_CUS", 0);
    e->split_tail = num("STDIFF_GEMM_SPLIT_TAIL", 0);
    e->splitk_max = num("STDIFF_SPLITK_MAX", 0);
    e->splitk_blocks = num("STDIFF_SPLITK_BLOCKS", 0);
    e->splitk_min = num("STDIFF_SPLITK_MIN", 0);
    // the LDS-DMA variant of the 128 x 128 kernel: 1 (default) for row-major W -- the forward GEMMs of a training step,
    // 16.4 -> 12.5 us per K = 1024 launch --, 2 also for K-major W (input gradients: faster alone, slower in the training
    // step, where its 128 KiB of LDS keep the weight-gradient group off the CU: 6.15 against 5.95 ms), 0 off
    e->dma128 = num("STDIFF_GEMM_DMA128", 1);
    e->depth = num("STDIFF_GEMM_DEPTH", 0);      // 1 / 3: force the 128 x 128 kernel's prefetch depth (A/B timing, tests)
    return e;
}
const GemmEnv* g_gemm_env = nullptr;
const GemmEnv& gemm_env() {
    const GemmEnv* e = __atomic_load_n(&g_gemm_env, __ATOMIC_ACQUIRE);
    if (!e) {
        const GemmEnv* fresh = read_gemm_env();
        const GemmEnv* expected = nullptr;
        if (__atomic_compare_exchange_n(&g_gemm_env, &expected, fresh, false, __ATOMIC_ACQ_REL, __ATOMIC_ACQUIRE)) e = fresh;
        else { delete fresh; e = expected; }
    }
    return *e;
}

constexpr int MAX_DEVICES = 64;
void* g_scratch_ptr[MAX_DEVICES];          // device address of `scratch` on each GPU (0 = not looked up yet)

// device address of this GPU's copy of `scratch`; looked up once per device with that device current
int static_scratch(void** out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    if (dev < 0 || dev >= MAX_DEVICES) return STDIFF_ERR_ARG;
    void* p = __atomic_load_n(&g_scratch_ptr[dev], __ATOMIC_ACQUIRE);
    if (!p) {
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return (int)e;
        if (!p) return STDIFF_ERR_STATE;
        __atomic_store_n(&g_scratch_ptr[dev], p, __ATOMIC_RELEASE);      // every thread finds the same address
    }
    *out = p;
    return STDIFF_OK;
}
}  // namespace

extern "C" void stdiff_reload_env(void) { __atomic_store_n(&g_gemm_env, read_gemm_env(), __ATOMIC_RELEASE); }

extern "C" int stdiff_gemm(const stdiff_gemm_args* a, void* stream) {
    STDIFF_CHECK_ARG(a != nullptr);
    return gemm_impl(a, stream, a->workspace, a->workspace_bytes);
}

// stdiff_gemm for callers that cannot allocate: the split-K partial products go to the library's own static device
// array.  No hipMalloc / hipFree / allocating routine is called, here or below.
extern "C" int stdiff_gemm_static_scratch(const stdiff_gemm_args* a, void* stream) {
    STDIFF_CHECK_ARG(a != nullptr && a->M >= 0 && a->N >= 0);
    void* ws = nullptr;
    const int rc = static_scratch(&ws);
    if (rc != STDIFF_OK) return rc;
    // strict: a shape whose K axis WOULD be split (few tiles, long K) but whose smallest split (STDIFF_SPLITK_MIN = 3
    // slices of the [M, N] fp32 partial product) does not fit the array is rejected with STDIFF_ERR_SCRATCH; shapes
    // that never split (every large GEMM of the sampler) need no scratch and pass through whatever their size
    return gemm_impl(a, stream, ws, (long long)STDIFF_STATIC_SCRATCH_BYTES, true);
}

namespace {
int gemm_impl(const stdiff_gemm_args* a, void* stream, void* workspace, long long workspace_bytes, bool strict_ws) {
    STDIFF_CHECK_ARG(a && a->M >= 0 && a->N >= 0 && a->K > 0);
    if (a->defer) a->defer->slices = 0;
    if (a->M == 0 || a->N == 0) return STDIFF_OK;          // an empty shard: nothing to do (pointers may be null)
    STDIFF_CHECK_ARG(a->A && a->W && a->C);
    STDIFF_CHECK_ARG(a->a_dtype == STDIFF_F32 || a->a_dtype == STDIFF_BF16);
    STDIFF_CHECK_ARG(a->c_dtype == STDIFF_F32 || a->c_dtype == STDIFF_BF16);
    STDIFF_CHECK_ARG(a->epi >= STDIFF_EPI_BIAS && a->epi <= STDIFF_EPI_GELU_BWD);
    if (a->M == 0 || a->N == 0) return STDIFF_OK;
    const bool bf = a->a_dtype == STDIFF_BF16;
    const int bk = bf ? 64 : 16, epv = bf ? 8 : 4;
    if (a->K % bk) return STDIFF_ERR_SHAPE;
    if (a->lda % epv || a->ldw % epv || !al16(a->A) || !al16(a->W)) return STDIFF_ERR_ALIGN;
    const bool ak = a->a_layout == STDIFF_K_MAJOR, wk = a->w_layout == STDIFF_K_MAJOR;
    STDIFF_CHECK_ARG((a->a_layout == STDIFF_ROW_MAJOR || ak) && (a->w_layout == STDIFF_ROW_MAJOR || wk));
    const int k_valid = a->k_valid > 0 ? a->k_valid : a->K;
    STDIFF_CHECK_ARG(k_valid <= a->K && (k_valid == a->K || ak || wk));
    // K-major operands: [k_valid][M] resp. [k_valid][N], 16-byte column pieces (M resp. N a multiple of 8 / 4)
    if ((ak && a->M % epv) || (wk && a->N % epv)) return STDIFF_ERR_SHAPE;
    if ((ak ? a->lda < a->M : a->lda < a->K) || (wk ? a->ldw < a->N : a->ldw < a->K) ||
        a->ldc < (a->epi == STDIFF_EPI_LN_MOD ? a->N / 2 : a->N))
        return STDIFF_ERR_ARG;
    if (a->epi == STDIFF_EPI_GATE_RES) STDIFF_CHECK_ARG(a->gate && a->res && a->c_dtype == STDIFF_F32);
    if (a->epi == STDIFF_EPI_ADD) STDIFF_CHECK_ARG(a->res != nullptr);
    if (a->epi == STDIFF_EPI_GELU_BWD) STDIFF_CHECK_ARG(a->gate != nullptr && a->ldgate >= a->N && !a->C2);
    if (a->epi == STDIFF_EPI_ACCUM) STDIFF_CHECK_ARG(a->c_dtype == STDIFF_F32);
    if (a->epi == STDIFF_EPI_GELU_SAVE) STDIFF_CHECK_ARG(a->C2 != nullptr);
    if (a->C2) STDIFF_CHECK_ARG(a->ldc2 >= a->N);
    const bool lnmod = a->epi == STDIFF_EPI_LN_MOD;
    if (lnmod) {       // C is [M, N/2]; persistent 256 x 256 kernel only (its B0 / B1 halves are the shift / scale rows)
        STDIFF_CHECK_ARG(a->res && a->ln_parts && a->ln_nparts >= 1 && a->ln_nparts <= 8 && (a->N / 2) % a->ln_nparts == 0 && !a->C2 && !ak && !wk);
        STDIFF_CHECK_ARG(a->a_dtype == STDIFF_BF16 && a->c_dtype == STDIFF_BF16 && a->ldc >= a->N / 2 && a->ldres >= a->N / 2);
        if (a->N % 256) return STDIFF_ERR_SHAPE;
        if (a->ldc % 8 || a->ldres % 4 || !al16(a->C) || !al16(a->res) || (a->bias && !al16(a->bias))) return STDIFF_ERR_ALIGN;
    }

    Params p;
    p.M = a->M; p.N = a->N; p.K = a->K;
    p.A = a->A; p.lda = a->lda; p.W = a->W; p.ldw = a->ldw; p.bias = a->bias;
    p.C = a->C; p.ldc = a->ldc; p.c_bf16 = a->c_dtype == STDIFF_BF16;
    p.epi = a->epi; p.gate = a->gate; p.ldgate = a->ldgate; p.res = a->res; p.ldres = a->ldres;
    p.C2 = a->C2; p.ldc2 = a->ldc2;
    p.ln_parts = a->ln_parts; p.ln_nparts = a->ln_nparts; p.stats_out = a->stats_out; p.colsum = nullptr; p.sumsq = nullptr;
    // 4-wide pieces: fp32 needs 16-byte, bf16 8-byte alignment of every row piece
    bool vec = a->N % 8 == 0 && a->ldc % 4 == 0 && (p.c_bf16 ? al8(a->C) : al16(a->C)) && (!a->bias || al16(a->bias));
    if (a->epi == STDIFF_EPI_GATE_RES)
        vec = vec && a->ldgate % 4 == 0 && (bf ? al8(a->gate) : al16(a->gate)) && al16(a->res) && a->ldres % 4 == 0;
    if (a->epi == STDIFF_EPI_GELU_BWD) vec = vec && a->ldgate % 4 == 0 && (bf ? al8(a->gate) : al16(a->gate));
    if (a->epi == STDIFF_EPI_ADD) vec = vec && al16(a->res) && a->ldres % 4 == 0;
    if (a->C2) vec = vec && a->ldc2 % 4 == 0 && (bf ? al8(a->C2) : al16(a->C2));
    p.vec = vec ? ((p.c_bf16 && a->ldc % 8 == 0 && al16(a->C)) ? 2 : 1) : 0;   // 2: bf16 C rows allow 16-byte stores

    // tile choice (bf16): the 256 x 256 pipelined kernel (1 block/CU) when its grid keeps the chip busy
    const GemmEnv& env = gemm_env();
    const int forced = env.tile;                         // 1 / 3 / 4 force a kernel (tests, A/B timing)
    // Measured at M = 20000, K = 1024 (profiles/): the 256x256 kernel delivers ~1.43x the per-output rate of the
    // 128x128 one, but runs ONE block per CU, so its grid is quantised in rounds of 256 tiles; the 128x128 kernel
    // (2 blocks per CU, short blocks) degrades smoothly.
    int tile = 1;
    if (bf && a->M > 128 && a->N > 128) {
        const double b256 = (double)cdiv(a->M, 256) * cdiv(a->N, 256), b128 = (double)cdiv(a->M, 128) * cdiv(a->N, 128);
        const double c256 = ceil(b256 / 256.0) * 2.0, c128 = fmax(1.0, b128 / 512.0) / (a->K >= 2048 ? 0.60 : 0.70);
        // with >= 128 tiles the persistent kernel is preferred outright: concurrent lanes fill its partial last round
        tile = (b256 >= 128.0 || c256 <= c128) ? 3 : 1;
    }
    if (bf && (forced == 1 || forced == 3)) tile = forced;
    // two 256 x 256 kernels: gemm256q (two 32-MFMA phases per slab) is the default, STDIFF_GEMM_TILE=3 selects the
    // four-phase gemm256p it was derived from (A/B timing), 4 forces the default kernel for any shape
    bool variant_q = forced != 3 && ((double)a->M + 256) * a->lda < 2.0e9 && ((double)a->N + 256) * a->ldw < 2.0e9;   // 32-bit DMA offsets
    if (bf && forced == 4) tile = 3;
    // two tiles in flight (gemm192t_kernel): 16-bit output with bias / GELU only, whole 16-byte C rows, a K loop long
    // enough to carry the previous tile out (13 slabs).  OPT-IN (STDIFF_GEMM_TILE=5): measured at the sampler's shapes it
    // runs at 0.69-0.70 of gemm256q's rate (profiles/notes/r03_gemm192t.md) -- the smaller tile's K loop loses more than
    // the hidden epilogue wins -- so no heuristic selects it; it stays as a second implementation for tests and A/B.
    const bool t192 = forced == 5 && bf && !ak && !wk && p.vec == 2 && p.c_bf16 && !a->C2 && !lnmod && !a->stats_out &&
                      (a->epi == STDIFF_EPI_BIAS || a->epi == STDIFF_EPI_GELU) && a->K / 64 >= 13 && variant_q;
    if (a->stats_out) {        // produced by the gate-residual tile epilogue of the persistent kernel only
        STDIFF_CHECK_ARG(a->epi == STDIFF_EPI_GATE_RES && bf && !a->C2 && !ak && !wk);
        if (a->N % 256) return STDIFF_ERR_SHAPE;
        if (!vec) return STDIFF_ERR_ALIGN;
    }
    if (lnmod || a->stats_out) {
        if (!variant_q || forced == 3) return STDIFF_ERR_SHAPE;     // operands beyond the 32-bit offsets of the LDS-DMA kernel
        tile = 3;
    }
    p.k_valid = k_valid;
    if (ak || wk) tile = 1;                       // K-major staging exists in the 128 x 128 kernel only
    if (a->epi == STDIFF_EPI_GELU_BWD) tile = 1;  // its rounding of v lives in the 128 x 128 kernel's epilogue (a training GEMM)
    if (t192) tile = 5;
    p.tiles_m = cdiv(a->M, tile == 1 ? BM : (tile == 5 ? 192 : 256));
    p.tiles_n = cdiv(a->N, tile == 1 ? BN : 256);
    p.ksplit = 1; p.nk_per = a->K / bk; p.ws = nullptr; p.ldws = 0;
    if (tile == 1 && workspace && ((uintptr_t)workspace & 15) == 0) {
        // few tiles x long K (batch-sized training GEMMs): one block per tile would walk K alone on a mostly idle chip
        const long tiles = (long)p.tiles_m * p.tiles_n;
        const int nk = a->K / bk, ldws = (a->N + 3) & ~3;
        // Target: ONE block per CU.  Measured on the 512-cell training step (STDIFF_SPLITK_BLOCKS = 128 ... 512): 256
        // blocks give 3.71 ms (bf16) / 7.30 ms (fp32) against 3.82 / 7.81 for 384 and 3.99 for 512 -- more slices are
        // more partial-product traffic for the reduce kernel than the second block per CU wins back.
        const int smax = env.splitk_max > 0 ? env.splitk_max : 16;
        const int sblocks = env.splitk_blocks > 0 ? env.splitk_blocks : cu_count();
        int S = (int)fmin(fmin((double)smax, nk / 4.0), (double)sblocks / tiles);
        const long long per = (long long)a->M * ldws * 4;
        // a 2-way split does not pay for its reduce launch (3.39 -> 3.35 ms per 512-cell step with 3 as the minimum)
        const int smin = env.splitk_min > 1 ? env.splitk_min : 3;
        if (S > 1 && per * S > workspace_bytes) {
            if (strict_ws && S >= smin && per * smin > workspace_bytes) return STDIFF_ERR_SCRATCH;
            S = (int)(workspace_bytes / per);
        }
        if (S >= smin) {
            p.nk_per = cdiv(nk, S);
            p.ksplit = cdiv(nk, p.nk_per);
            p.ws = (float*)workspace; p.ldws = ldws;
        }
    }
    dim3 grid(p.tiles_m * p.tiles_n * p.ksplit), block(tile == 1 ? NT : 512);
    p.t_full = p.units = p.tiles_m * p.tiles_n;
    if (tile == 5) {
        int cus = cu_count();
        if (env.cus > 0 && env.cus < cus) cus = env.cus;
        grid.x = min((int)grid.x, cus);
    }
    if (tile == 3) {                                           // persistent: one workgroup per CU walks the tiles
        int cus = cu_count();
        if (env.cus > 0 && env.cus < cus) cus = env.cus;
        // STDIFF_GEMM_SPLIT_TAIL=1 (opt-in): a last round that fills at most half of the CUs is cut into 128-row half
        // units (two-phase kernel only).  Measured: one stream gains 2.6 % over the 33 launches (fc2 222 -> 209 us), but
        // with two concurrent lanes the step is 2.3 % SLOWER -- the CUs a short round leaves free are exactly where the
        // other lane's kernels run -- so the default stays off.
        const int T = p.units, frac = T % cus;
        if (variant_q && frac > 0 && 2 * frac <= cus && env.split_tail == 1) {
            p.t_full = T - frac;
            p.units = p.t_full + 2 * frac;
            grid.x = p.units;
        }
        grid.x = min((int)grid.x, cus);
    }
    // few blocks (at most one per CU): nothing else on the CU hides the operand latency -> three slabs in flight
    const int depth_env = env.depth;
    const bool deep = tile == 1 && (depth_env == 3 || (depth_env == 0 && (long)grid.x <= (long)cu_count()));
#define STDIFF_LAUNCH128_D(T, D)                                                                               \
    do {                                                                                                       \
        if (ak && wk) hipLaunchKernelGGL((gemm_nt_kernel<T, true, true, D>), grid, block, 0, (hipStream_t)stream, p);     \
        else if (ak) hipLaunchKernelGGL((gemm_nt_kernel<T, true, false, D>), grid, block, 0, (hipStream_t)stream, p);     \
        else if (wk) hipLaunchKernelGGL((gemm_nt_kernel<T, false, true, D>), grid, block, 0, (hipStream_t)stream, p);     \
        else hipLaunchKernelGGL((gemm_nt_kernel<T, false, false, D>), grid, block, 0, (hipStream_t)stream, p);            \
    } while (0)
#define STDIFF_LAUNCH128(T)                                                                                    \
    do {                                                                                                       \
        if (deep) STDIFF_LAUNCH128_D(T, 3);                                                                    \
        else STDIFF_LAUNCH128_D(T, 1);                                                                         \
    } while (0)
    // 32-bit buffer offsets: both operands below 2 GiB
    const bool dma128 = bf && tile == 1 && !ak && deep && env.dma128 != 0 && (!wk || env.dma128 == 2) &&
                        ((unsigned long long)(a->M - 1) * a->lda + a->K) * 2ull < 0x7fffffffull &&
                        (wk ? ((unsigned long long)(k_valid - 1) * a->ldw + a->N)
                            : ((unsigned long long)(a->N - 1) * a->ldw + a->K)) * 2ull < 0x7fffffffull;
    static const int dma_waves = [] { const char* e = getenv("STDIFF_GEMM_DMA128_WAVES"); return e && atoi(e) == 4 ? 4 : 8; }();
    if (dma128 && wk && dma_waves == 8)
        hipLaunchKernelGGL((gemm128d_kernel<true, 8>), grid, dim3(512), 0, (hipStream_t)stream, p);
    else if (dma128 && wk)
        hipLaunchKernelGGL((gemm128d_kernel<true, 4>), grid, block, 0, (hipStream_t)stream, p);
    else if (dma128 && dma_waves == 8)
        hipLaunchKernelGGL((gemm128d_kernel<false, 8>), grid, dim3(512), 0, (hipStream_t)stream, p);
    else if (dma128)
        hipLaunchKernelGGL((gemm128d_kernel<false, 4>), grid, block, 0, (hipStream_t)stream, p);
    else if (!bf)
        STDIFF_LAUNCH128(float);
    else if (tile == 1)
        STDIFF_LAUNCH128(bf16_t);
    else if (tile == 5 && a->epi == STDIFF_EPI_GELU)
        hipLaunchKernelGGL(gemm192t_kernel<STDIFF_EPI_GELU>, grid, block, 0, (hipStream_t)stream, p);
    else if (tile == 5)
        hipLaunchKernelGGL(gemm192t_kernel<STDIFF_EPI_BIAS>, grid, block, 0, (hipStream_t)stream, p);
    else if (variant_q && lnmod)
        hipLaunchKernelGGL(gemm256q_kernel<1>, grid, block, 0, (hipStream_t)stream, p);
    else if (variant_q && a->stats_out)
        hipLaunchKernelGGL(gemm256q_kernel<2>, grid, block, 0, (hipStream_t)stream, p);
    else if (variant_q)
        hipLaunchKernelGGL(gemm256q_kernel<0>, grid, block, 0, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL(gemm256p_kernel, grid, block, 0, (hipStream_t)stream, p);
    if (a->defer) a->defer->slices = 0;
    if (p.ksplit > 1 && a->defer && (a->epi == STDIFF_EPI_BIAS || a->epi == STDIFF_EPI_GATE_RES)) {
        // the consumer sums the slices (stdiff_*_split): no reduce launch
        a->defer->ws = p.ws; a->defer->ldws = p.ldws; a->defer->slices = p.ksplit;
        a->defer->slice_stride = (long long)p.M * p.ldws;
    } else if (p.ksplit > 1) {
        dim3 rg(cdiv((long)p.M * (p.ldws >> 2), 256));
        if (bf) hipLaunchKernelGGL(splitk_reduce_kernel<bf16_t>, rg, dim3(256), 0, (hipStream_t)stream, p);
        else hipLaunchKernelGGL(splitk_reduce_kernel<float>, rg, dim3(256), 0, (hipStream_t)stream, p);
    }
    STDIFF_LAUNCH_STATUS();
    return STDIFF_OK;
}
}  // namespace

// floats of stdiff_dw_group_args.sq_parts the launch writes: one per block of the grid before any cap
extern "C" long long stdiff_dw_group_parts(const stdiff_dw_group_args* a) {
    if (!a || a->n < 1 || a->n > STDIFF_DW_GROUP_MAX) return -1;
    long long end = 0;
    for (int i = 0; i < a->n; ++i) {
        if (a->p[i].M <= 0 || a->p[i].N <= 0) return -1;
        end += ((long long)cdiv(a->p[i].M, BM) * cdiv(a->p[i].N, BN) + 7) & ~7LL;
    }
    return end;
}

// The weight gradients of several linear layers over one batch, one launch (gemm_dw_group_kernel).
extern "C" int stdiff_dw_group(const stdiff_dw_group_args* a, void* stream) {
    STDIFF_CHECK_ARG(a != nullptr && a->n >= 1 && a->n <= STDIFF_DW_GROUP_MAX && a->C >= 0);
    STDIFF_CHECK_ARG(a->dtype == STDIFF_F32 || a->dtype == STDIFF_BF16);
    if (a->sq_parts && ((uintptr_t)a->sq_parts & 3)) return STDIFF_ERR_ALIGN;
    if (a->C == 0) {
        if (a->sq_parts) {
            const long long n = stdiff_dw_group_parts(a);
            STDIFF_CHECK_ARG(n >= 0);
            const hipError_t e = hipMemsetAsync(a->sq_parts, 0, (size_t)n * 4, (hipStream_t)stream);
            if (e != hipSuccess) return (int)e;
        }
        return STDIFF_OK;
    }
    const bool bf = a->dtype == STDIFF_BF16;
    const int bk = bf ? 64 : 16, epv = bf ? 8 : 4;
    DwGroup g;
    g.n = a->n; g.k_valid = a->C; g.K = (a->C + bk - 1) / bk * bk; g.sumsq = a->sq_parts;
    int end = 0;
    for (int i = 0; i < a->n; ++i) {
        const stdiff_dw_problem& q = a->p[i];
        STDIFF_CHECK_ARG(q.M > 0 && q.N > 0 && q.dY && q.X && q.gw && q.lddy >= q.M && q.ldx >= q.N && q.ldgw >= q.N);
        if (q.M % epv || q.N % 8) return STDIFF_ERR_SHAPE;          // 16-byte column pieces; 8-column epilogue pieces
        if (q.lddy % epv || q.ldx % epv || q.ldgw % 4 || !al16(q.dY) || !al16(q.X) || !al16(q.gw)) return STDIFF_ERR_ALIGN;
        DwProblem& d = g.pr[i];
        d.M = q.M; d.N = q.N; d.A = q.dY; d.lda = q.lddy; d.W = q.X; d.ldw = q.ldx; d.C = q.gw; d.ldc = q.ldgw;
        d.colsum = q.gb;
        d.tiles_m = cdiv(q.M, BM); d.tiles_n = cdiv(q.N, BN);
        end += (d.tiles_m * d.tiles_n + 7) & ~7;
        d.blk_end = end;
    }
    for (int i = a->n; i < DW_MAX; ++i) g.pr[i] = DwProblem{};
    g.total = end;
    // grid: one block per tile.  STDIFF_DW_GROUP_BLOCKS = n > 0 (read once) caps it (the blocks then walk the tiles):
    // measured on the training iteration, one resident block per CU -- leaving the dX chain a slot on every CU -- is
    // SLOWER (2.54 against 2.46 ms at H = 1024, 6.56 against 6.15 at configs[2]'s shape); 512 blocks equal the full grid
    static const int gblocks = [] { const char* e = getenv("STDIFF_DW_GROUP_BLOCKS"); return e ? atoi(e) : 0; }();
    const int grid = gblocks > 0 ? min(end, gblocks) : end;
    if (bf) hipLaunchKernelGGL(gemm_dw_group_kernel<bf16_t>, dim3(grid), dim3(NT), 0, (hipStream_t)stream, g);
    else hipLaunchKernelGGL(gemm_dw_group_kernel<float>, dim3(grid), dim3(NT), 0, (hipStream_t)stream, g);
    STDIFF_LAUNCH_STATUS();
    return STDIFF_OK;
}

#ifdef STDIFF_T192_STAMPS
extern "C" int stdiff_debug_t192_stamps(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_t192), sizeof(g_t192));
}
#endif

#ifdef STDIFF_GEMM_STAMPS
extern "C" int stdiff_debug_gemm_stamps(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_stamps), sizeof(g_stamps));
}
#endif

extern "C" int stdiff_transpose(const void* in, int ld_in, void* out, int ld_out, int rows, int cols, int rows_pad,
                                int dtype, void* stream) {
    STDIFF_CHECK_ARG(in && out && rows >= 0 && cols >= 0 && ld_in >= cols && rows_pad >= rows && ld_out >= rows_pad);
    STDIFF_CHECK_ARG(dtype == STDIFF_F32 || dtype == STDIFF_BF16);
    if (rows == 0 || cols == 0) return STDIFF_OK;
    dim3 grid(cdiv(cols, 64), cdiv(rows_pad, 64)), block(256);
    if (dtype == STDIFF_BF16)
        hipLaunchKernelGGL(transpose_kernel<bf16_t>, grid, block, 0, (hipStream_t)stream, (const bf16_t*)in, ld_in,
                           (bf16_t*)out, ld_out, rows, cols, rows_pad);
    else
        hipLaunchKernelGGL(transpose_kernel<float>, grid, block, 0, (hipStream_t)stream, (const float*)in, ld_in,
                           (float*)out, ld_out, rows, cols, rows_pad);
    STDIFF_LAUNCH_STATUS();
    return STDIFF_OK;
}
