// NT GEMM on the gfx950 matrix cores:  C[M,N] = epi(A[M,K] . W[N,K]^T + bias[N]).
//
// Every dense contraction of the denoiser (nn.Linear: stDiff_model.py:18,20,54,56,107,128,132,177,180 and the
// timm Mlp fc1/fc2) is this one shape: both operands are K-contiguous, so both tiles are staged as [rows][K-slab]
// and both MFMA fragments are plain 16-byte LDS reads.
//
//   block  = 256 threads = 4 waves (2 x 2), block tile 128 x 128, wave tile 64 x 64 = 4 x 4 MFMA tiles of 16 x 16
//   bf16   : K-slab 64,  v_mfma_f32_16x16x32_bf16, fp32 accumulate
//   f32    : K-slab 16,  v_mfma_f32_16x16x4_f32  (exact fp32: a k-ordered fmaf chain)
//   LDS    : 2 x (A tile + W tile), 16-byte slots XOR-swizzled so ds_read_b128 is conflict-free
//            (tools/lds_conflicts.py); global->register->LDS staging, next slab's loads in flight under the MFMAs
//   MFMA is issued "swapped" (W fragment as the A operand) so a lane ends up holding 4 CONSECUTIVE n of one m:
//   the accumulators go to an fp32 LDS image of the C tile with one ds_write_b128 each, and the epilogue
//   (bias / SiLU / GELU / gate*v+res / ...) then runs on whole 32-byte row pieces with coalesced global traffic.
//   grid   : 1-D, XCD-aware (blocks b and b+8 share an L2): each XCD walks a contiguous range of tiles, ordered
//            in groups of 8 row-tiles x all column-tiles so the A panel and the weight panel stay L2-resident.
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
    int vec;  // all epilogue pointers/strides allow 8-wide vector access
    int tiles_m, tiles_n;
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

template <typename T>
__device__ __forceinline__ void epilogue8(const Params& p, int gm, int gn, const float* v) {
    float o[8];
    float pre[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) pre[j] = v[j];
    if (p.bias) {
        if (p.vec) {
            f32x4 b0 = *(const f32x4*)(p.bias + gn), b1 = *(const f32x4*)(p.bias + gn + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) { pre[j] += b0[j]; pre[4 + j] += b1[j]; }
        } else {
            for (int j = 0; j < 8; ++j) if (gn + j < p.N) pre[j] += p.bias[gn + j];
        }
    }
    switch (p.epi) {
        case STDIFF_EPI_SILU:
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = silu_f(pre[j]);
            break;
        case STDIFF_EPI_GELU:
        case STDIFF_EPI_GELU_SAVE:
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = gelu_erf_f(pre[j]);
            break;
        case STDIFF_EPI_GATE_RES: {
            const T* g = (const T*)p.gate + (size_t)gm * p.ldgate + gn;
            const float* r = p.res + (size_t)gm * p.ldres + gn;
            if (p.vec) {
                f32x4 g0 = ld4(g), g1 = ld4(g + 4), r0 = *(const f32x4*)r, r1 = *(const f32x4*)(r + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { o[j] = r0[j] + g0[j] * pre[j]; o[4 + j] = r1[j] + g1[j] * pre[4 + j]; }
            } else {
                for (int j = 0; j < 8; ++j) o[j] = (gn + j < p.N) ? r[j] + Elem<T>::ld(g + j) * pre[j] : 0.f;
            }
            break;
        }
        case STDIFF_EPI_ADD: {
            const float* r = p.res + (size_t)gm * p.ldres + gn;
            if (p.vec) {
                f32x4 r0 = *(const f32x4*)r, r1 = *(const f32x4*)(r + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { o[j] = pre[j] + r0[j]; o[4 + j] = pre[4 + j] + r1[j]; }
            } else {
                for (int j = 0; j < 8; ++j) o[j] = (gn + j < p.N) ? pre[j] + r[j] : 0.f;
            }
            break;
        }
        case STDIFF_EPI_ACCUM: {
            const float* r = (const float*)p.C + (size_t)gm * p.ldc + gn;
            if (p.vec) {
                f32x4 r0 = *(const f32x4*)r, r1 = *(const f32x4*)(r + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { o[j] = pre[j] + r0[j]; o[4 + j] = pre[4 + j] + r1[j]; }
            } else {
                for (int j = 0; j < 8; ++j) o[j] = (gn + j < p.N) ? pre[j] + r[j] : 0.f;
            }
            break;
        }
        default:
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = pre[j];
    }
    auto store = [&](void* base, int ld, const float* w, bool bf) {
        if (bf) {
            bf16_t* c = (bf16_t*)base + (size_t)gm * ld + gn;
            if (p.vec) {
                u32x4 pk;
#pragma unroll
                for (int j = 0; j < 4; ++j) pk[j] = (unsigned)f2bf(w[2 * j]) | ((unsigned)f2bf(w[2 * j + 1]) << 16);
                *(u32x4*)c = pk;
            } else {
                for (int j = 0; j < 8; ++j) if (gn + j < p.N) c[j] = f2bf(w[j]);
            }
        } else {
            float* c = (float*)base + (size_t)gm * ld + gn;
            if (p.vec) {
                *(f32x4*)c = f32x4{w[0], w[1], w[2], w[3]};
                *(f32x4*)(c + 4) = f32x4{w[4], w[5], w[6], w[7]};
            } else {
                for (int j = 0; j < 8; ++j) if (gn + j < p.N) c[j] = w[j];
            }
        }
    };
    store(p.C, p.ldc, o, p.c_bf16);
    if (p.C2) store(p.C2, p.ldc2, pre, sizeof(T) == 2);   // pre-activation v, kept in the operand dtype
}

template <typename T>
__global__ __launch_bounds__(NT, 2) void gemm_nt_kernel(Params p) {
    using C = Cfg<T>;
    constexpr int BK = C::BK, EPV = C::EPV, NSLOT = C::NSLOT, ROWB = C::ROWB;
    constexpr int TILEB = BM * ROWB;               // bytes of one operand tile
    constexpr int LPT = BM * NSLOT / NT;           // 16-byte loads per thread per operand
    constexpr int STAGE = 4 * TILEB;               // 2 buffers x (A + W)
    constexpr int CBYTES = BM * CLD * 4;
    constexpr int SMEM = STAGE > CBYTES ? STAGE : CBYTES;
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];

    int tm, tn;
    tile_of(blockIdx.x, gridDim.x, p.tiles_m, p.tiles_n, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid >> 1, wn = wid & 1;

    // ---- staging addresses (rows past the edge are clamped: their products are never stored)
    const T* ga[LPT];
    const T* gw[LPT];
    int soff[LPT];
#pragma unroll
    for (int i = 0; i < LPT; ++i) {
        int c = tid + NT * i;
        int row = c / NSLOT, slot = c % NSLOT;
        ga[i] = (const T*)p.A + (size_t)min(m0 + row, p.M - 1) * p.lda + slot * EPV;
        gw[i] = (const T*)p.W + (size_t)min(n0 + row, p.N - 1) * p.ldw + slot * EPV;
        soff[i] = row * ROWB + ((slot ^ C::swz(row)) << 4);
    }
    u32x4 ra[LPT], rw[LPT];
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            ra[i] = *(const u32x4*)(ga[i] + k0);
            rw[i] = *(const u32x4*)(gw[i] + k0);
        }
    };
    auto sstore = [&](int buf) {
        unsigned char* b = smem + buf * 2 * TILEB;
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            *(u32x4*)(b + soff[i]) = ra[i];
            *(u32x4*)(b + TILEB + soff[i]) = rw[i];
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fq = lane >> 4;
    auto compute = [&](int buf) {
        const unsigned char* ba = smem + buf * 2 * TILEB;
        const unsigned char* bw = ba + TILEB;
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 af[4], wf[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    int ra_ = wm * 64 + t * 16 + fr, rw_ = wn * 64 + t * 16 + fr;
                    int slot = ks * 4 + fq;
                    af[t] = *(const bf16x8*)(ba + ra_ * ROWB + ((slot ^ C::swz(ra_)) << 4));
                    wf[t] = *(const bf16x8*)(bw + rw_ * ROWB + ((slot ^ C::swz(rw_)) << 4));
                }
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt], af[mt], acc[mt][nt], 0, 0, 0);
            }
        } else {
            f32x4 af[4], wf[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                int ra_ = wm * 64 + t * 16 + fr, rw_ = wn * 64 + t * 16 + fr;
                af[t] = *(const f32x4*)(ba + ra_ * ROWB + ((fq ^ C::swz(ra_)) << 4));
                wf[t] = *(const f32x4*)(bw + rw_ * ROWB + ((fq ^ C::swz(rw_)) << 4));
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[nt][kk], af[mt][kk], acc[mt][nt], 0, 0, 0);
        }
    };

    const int nk = p.K / BK;
    gload(0);
    sstore(0);
    __syncthreads();
    for (int k = 0; k < nk; ++k) {
        if (k + 1 < nk) gload((k + 1) * BK);
        compute(k & 1);
        if (k + 1 < nk) sstore((k + 1) & 1);
        __syncthreads();
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
    const int c8 = (tid & 15) * 8;
#pragma unroll
    for (int ps = 0; ps < 8; ++ps) {
        int r = (tid >> 4) + 16 * ps;
        int gm = m0 + r, gn = n0 + c8;
        if (gm < p.M && gn < p.N) {
            f32x4 v0 = *(const f32x4*)(cl + r * CLD + c8), v1 = *(const f32x4*)(cl + r * CLD + c8 + 4);
            float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            epilogue8<T>(p, gm, gn, v);
        }
    }
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

}  // namespace

extern "C" int stdiff_gemm(const stdiff_gemm_args* a, void* stream) {
    STDIFF_CHECK_ARG(a && a->A && a->W && a->C);
    STDIFF_CHECK_ARG(a->M >= 0 && a->N >= 0 && a->K > 0);
    STDIFF_CHECK_ARG(a->a_dtype == STDIFF_F32 || a->a_dtype == STDIFF_BF16);
    STDIFF_CHECK_ARG(a->c_dtype == STDIFF_F32 || a->c_dtype == STDIFF_BF16);
    STDIFF_CHECK_ARG(a->epi >= STDIFF_EPI_BIAS && a->epi <= STDIFF_EPI_ACCUM);
    if (a->M == 0 || a->N == 0) return STDIFF_OK;
    const bool bf = a->a_dtype == STDIFF_BF16;
    const int bk = bf ? 64 : 16, epv = bf ? 8 : 4;
    if (a->K % bk) return STDIFF_ERR_SHAPE;
    if (a->lda % epv || a->ldw % epv || !al16(a->A) || !al16(a->W)) return STDIFF_ERR_ALIGN;
    if (a->lda < a->K || a->ldw < a->K || a->ldc < a->N) return STDIFF_ERR_ARG;
    if (a->epi == STDIFF_EPI_GATE_RES) STDIFF_CHECK_ARG(a->gate && a->res && a->c_dtype == STDIFF_F32);
    if (a->epi == STDIFF_EPI_ADD) STDIFF_CHECK_ARG(a->res != nullptr);
    if (a->epi == STDIFF_EPI_ACCUM) STDIFF_CHECK_ARG(a->c_dtype == STDIFF_F32);
    if (a->epi == STDIFF_EPI_GELU_SAVE) STDIFF_CHECK_ARG(a->C2 != nullptr);

    Params p;
    p.M = a->M; p.N = a->N; p.K = a->K;
    p.A = a->A; p.lda = a->lda; p.W = a->W; p.ldw = a->ldw; p.bias = a->bias;
    p.C = a->C; p.ldc = a->ldc; p.c_bf16 = a->c_dtype == STDIFF_BF16;
    p.epi = a->epi; p.gate = a->gate; p.ldgate = a->ldgate; p.res = a->res; p.ldres = a->ldres;
    p.C2 = a->C2; p.ldc2 = a->ldc2;
    const int cel = p.c_bf16 ? 8 : 4;  // elements per 16 B of the output
    bool vec = a->N % 8 == 0 && al16(a->C) && a->ldc % cel == 0 && (!a->bias || al16(a->bias));
    if (a->epi == STDIFF_EPI_GATE_RES)
        vec = vec && al16(a->gate) && a->ldgate % epv == 0 && al16(a->res) && a->ldres % 4 == 0;
    if (a->epi == STDIFF_EPI_ADD) vec = vec && al16(a->res) && a->ldres % 4 == 0;
    if (a->C2) vec = vec && al16(a->C2) && a->ldc2 % epv == 0 && a->ldc2 >= a->N;
    p.vec = vec;
    p.tiles_m = cdiv(a->M, BM);
    p.tiles_n = cdiv(a->N, BN);
    dim3 grid(p.tiles_m * p.tiles_n), block(NT);
    if (bf)
        hipLaunchKernelGGL(gemm_nt_kernel<bf16_t>, grid, block, 0, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL(gemm_nt_kernel<float>, grid, block, 0, (hipStream_t)stream, p);
    STDIFF_LAUNCH_STATUS();
    return STDIFF_OK;
}

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
