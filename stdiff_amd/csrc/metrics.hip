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

}  // namespace

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
