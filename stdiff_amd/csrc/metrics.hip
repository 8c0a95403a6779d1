// Per-gene accuracy statistics of an imputed matrix against the measured one, on the device.
// The reference computes its metrics column by column on the host with pandas/scipy in float64 (test-stDiff.py:203-364);
// here one pass over the two [N, G] fp32 matrices leaves, per gene, the float64 moments from which Pearson, cosine,
// RMSE-after-zscore and SSIM-after-scale_max follow in closed form, and a second pass the Jensen-Shannon divergence
// of the sum-normalised columns.  Both passes are HBM-bound (8 B per element) and deterministic: rows are cut into
// `chunks` slabs, each (slab, 64-column block) workgroup writes ONE partial, the host adds the slabs in order.
#include "common.h"

namespace {

constexpr int NMOM = 7;   // sum x, sum y, sum xy, sum xx, sum yy, max x, max y

__global__ __launch_bounds__(256) void column_moments_kernel(const float* __restrict__ x, int ldx,
                                                             const float* __restrict__ y, int ldy, double* part, int N,
                                                             int G, int rows_per_chunk) {
    __shared__ double red[4][NMOM][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int r0 = blockIdx.y * rows_per_chunk, r1 = min(N, r0 + rows_per_chunk);
    double sx = 0, sy = 0, sxy = 0, sxx = 0, syy = 0, mx = -INFINITY, my = -INFINITY;
    if (c < G) {
        for (int r = r0 + w; r < r1; r += 4) {
            const double a = x[(size_t)r * ldx + c], b = y[(size_t)r * ldy + c];
            sx += a; sy += b; sxy += a * b; sxx += a * a; syy += b * b;
            mx = fmax(mx, a); my = fmax(my, b);
        }
    }
    double v[NMOM] = {sx, sy, sxy, sxx, syy, mx, my};
#pragma unroll
    for (int k = 0; k < NMOM; ++k) red[w][k][lane] = v[k];
    __syncthreads();
    if (w == 0 && c < G) {
#pragma unroll
        for (int k = 0; k < NMOM; ++k) {
            double t;
            if (k < 5) t = (red[0][k][lane] + red[1][k][lane]) + (red[2][k][lane] + red[3][k][lane]);
            else t = fmax(fmax(red[0][k][lane], red[1][k][lane]), fmax(red[2][k][lane], red[3][k][lane]));
            part[((size_t)blockIdx.y * NMOM + k) * G + c] = t;
        }
    }
}

// JS(p, q) = 0.5 KL(p || m) + 0.5 KL(q || m), p = x / sum x, q = y / sum y, m = (p + q) / 2, natural log; terms with a
// zero numerator are 0 (scipy.stats.entropy's rel_entr).  inv_sx / inv_sy: 1 / column sums (from the moments pass).
__global__ __launch_bounds__(256) void column_js_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ y,
                                                        int ldy, const double* __restrict__ inv_sx,
                                                        const double* __restrict__ inv_sy, double* part, int N, int G,
                                                        int rows_per_chunk) {
    __shared__ double red[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int r0 = blockIdx.y * rows_per_chunk, r1 = min(N, r0 + rows_per_chunk);
    double acc = 0;
    if (c < G) {
        const double ix = inv_sx[c], iy = inv_sy[c];
        for (int r = r0 + w; r < r1; r += 4) {
            const double p = x[(size_t)r * ldx + c] * ix, q = y[(size_t)r * ldy + c] * iy;
            const double m = 0.5 * (p + q);
            if (p > 0) acc += 0.5 * p * log(p / m);
            if (q > 0) acc += 0.5 * q * log(q / m);
        }
    }
    red[w][lane] = acc;
    __syncthreads();
    if (w == 0 && c < G) part[(size_t)blockIdx.y * G + c] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

// Spearman (CalculateMeteics.SPCC, test-stDiff.py:300-316: scipy.stats.spearmanr = Pearson of the average ranks).
// Ranks by counting instead of sorting: for element i of a column, lt = #{j: v_j < v_i}, le = #{j: v_j <= v_i}; the
// average rank of its tie group is (lt + le + 1) / 2, and centred on the mean rank (N + 1) / 2 it is (lt + le - N) / 2:
// integers, exact.  The columns arrive transposed ([G][ldt], one gene contiguous), a workgroup owns 1024 elements of one
// gene in registers and streams the whole column through LDS (broadcast reads), 2 compares + 2 carry-adds per pair per
// matrix: VALU-bound, N^2 G pairs (20 000 x 1000: 4e11 pairs).  One partial {sum rx'^2, sum ry'^2, sum rx' ry'} per
// (gene, block), added on the host in block order.
constexpr int RANK_I = 4;        // elements per thread
constexpr int RANK_J = 2048;     // column values per LDS tile

__global__ __launch_bounds__(256) void rank_corr_kernel(const float* __restrict__ xt, const float* __restrict__ yt,
                                                        size_t ldt, double* __restrict__ part, int N) {
    __shared__ __align__(16) float sx[RANK_J];
    __shared__ __align__(16) float sy[RANK_J];
    __shared__ double red[4][3];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const float* xc = xt + (size_t)blockIdx.y * ldt;
    const float* yc = yt + (size_t)blockIdx.y * ldt;
    const int i0 = blockIdx.x * (256 * RANK_I);
    float xi[RANK_I], yi[RANK_I];
    int ltx[RANK_I], lex[RANK_I], lty[RANK_I], ley[RANK_I];
#pragma unroll
    for (int k = 0; k < RANK_I; ++k) {
        const int i = i0 + k * 256 + tid;
        xi[k] = i < N ? xc[i] : 0.f;
        yi[k] = i < N ? yc[i] : 0.f;
        ltx[k] = lex[k] = lty[k] = ley[k] = 0;
    }
    for (int j0 = 0; j0 < N; j0 += RANK_J) {
        __syncthreads();
        for (int j = tid; j < RANK_J; j += 256) {       // past the end: NaN compares false both ways
            sx[j] = j0 + j < N ? xc[j0 + j] : NAN;
            sy[j] = j0 + j < N ? yc[j0 + j] : NAN;
        }
        __syncthreads();
        const int jn = min(RANK_J, (N - j0 + 3) & ~3);
        for (int j = 0; j < jn; j += 4) {
            const float4 a = *reinterpret_cast<const float4*>(&sx[j]);
            const float4 b = *reinterpret_cast<const float4*>(&sy[j]);
#pragma unroll
            for (int k = 0; k < RANK_I; ++k) {
                ltx[k] += (a.x < xi[k]) + (a.y < xi[k]) + (a.z < xi[k]) + (a.w < xi[k]);
                lex[k] += (a.x <= xi[k]) + (a.y <= xi[k]) + (a.z <= xi[k]) + (a.w <= xi[k]);
                lty[k] += (b.x < yi[k]) + (b.y < yi[k]) + (b.z < yi[k]) + (b.w < yi[k]);
                ley[k] += (b.x <= yi[k]) + (b.y <= yi[k]) + (b.z <= yi[k]) + (b.w <= yi[k]);
            }
        }
    }
    double sxx = 0, syy = 0, sxy = 0;
#pragma unroll
    for (int k = 0; k < RANK_I; ++k) {
        if (i0 + k * 256 + tid < N) {
            const double rx = 0.5 * (double)(ltx[k] + lex[k] - N), ry = 0.5 * (double)(lty[k] + ley[k] - N);
            sxx += rx * rx; syy += ry * ry; sxy += rx * ry;
        }
    }
    // fixed-order reduction: lanes by xor-butterfly (same tree every run), then the four waves in order
    for (int o = 32; o; o >>= 1) {
        sxx += __shfl_xor(sxx, o); syy += __shfl_xor(syy, o); sxy += __shfl_xor(sxy, o);
    }
    if (lane == 0) { red[w][0] = sxx; red[w][1] = syy; red[w][2] = sxy; }
    __syncthreads();
    if (tid < 3)
        part[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 3 + tid] =
            (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}

}  // namespace

extern "C" int stdiff_rank_corr(const float* xt, const float* yt, long long ldt, double* partials, int blocks, int N,
                                int G, void* stream) {
    STDIFF_CHECK_ARG(xt && yt && partials && N > 0 && G > 0 && ldt >= N && G <= 65535);
    STDIFF_CHECK_ARG(blocks == cdiv(N, 256 * RANK_I));
    hipLaunchKernelGGL(rank_corr_kernel, dim3(blocks, G), dim3(256), 0, (hipStream_t)stream, xt, yt, (size_t)ldt,
                       partials, N);
    STDIFF_LAUNCH_STATUS();
    return STDIFF_OK;
}

extern "C" int stdiff_column_moments(const float* x, int ldx, const float* y, int ldy, double* partials, int chunks,
                                     int N, int G, void* stream) {
    STDIFF_CHECK_ARG(x && y && partials && N > 0 && G > 0 && chunks > 0 && ldx >= G && ldy >= G);
    const int rpc = cdiv(N, chunks);
    hipLaunchKernelGGL(column_moments_kernel, dim3(cdiv(G, 64), chunks), dim3(256), 0, (hipStream_t)stream, x, ldx, y,
                       ldy, partials, N, G, rpc);
    STDIFF_LAUNCH_STATUS();
    return STDIFF_OK;
}

extern "C" int stdiff_column_js(const float* x, int ldx, const float* y, int ldy, const double* inv_sum_x,
                                const double* inv_sum_y, double* partials, int chunks, int N, int G, void* stream) {
    STDIFF_CHECK_ARG(x && y && inv_sum_x && inv_sum_y && partials && N > 0 && G > 0 && chunks > 0 && ldx >= G && ldy >= G);
    const int rpc = cdiv(N, chunks);
    hipLaunchKernelGGL(column_js_kernel, dim3(cdiv(G, 64), chunks), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy,
                       inv_sum_x, inv_sum_y, partials, N, G, rpc);
    STDIFF_LAUNCH_STATUS();
    return STDIFF_OK;
}
