/* libstdiff_hip.so -- C ABI of the MI355X (gfx950) stDiff hot path.
 *
 * The reference (fdu-wangfeilab/stDiff) has no FFI: its hot path is eager PyTorch behind four
 * Python callables (DiT_stDiff, NoiseScheduler, normal_train_stDiff, sample_stDiff).  Each entry
 * point below names the reference lines whose arithmetic it replaces.  The Python package
 * stdiff_amd/ keeps the reference signatures and calls these through ctypes with raw device
 * pointers (torch owns all memory) and the caller's hipStream_t.
 *
 * Conventions
 *   - every function returns int: 0 ok, <0 STDIFF_ERR_* (bad argument), >0 a hipError_t;
 *     nothing throws, nothing aborts, nothing allocates device memory unless it says so;
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*); no internal threads,
 *     no host synchronisation, so every entry point may be stream-captured into a hipGraph
 *     (stdiff_graph_*);
 *   - matrices are row-major with an explicit leading dimension in ELEMENTS;
 *   - dtype: STDIFF_F32 (exact-fp32 MFMA path) or STDIFF_BF16 (16-bit MFMA operands, fp32 accumulate).  The
 *     16-bit FORMAT is a property of the build: libstdiff_hip.so computes in bf16; libstdiff_hip_f16.so (same
 *     sources, same entry points, -DSTDIFF_H16_FP16) in IEEE fp16 with saturating conversions, for sampling
 *     (stdiff_h16_format() tells which one was loaded).
 */
#ifndef STDIFF_HIP_H
#define STDIFF_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STDIFF_ABI_VERSION 14

enum { STDIFF_F32 = 0, STDIFF_BF16 = 1 };

enum {
    STDIFF_OK = 0,
    STDIFF_ERR_ARG = -1,      /* null pointer / negative size / bad enum */
    STDIFF_ERR_SHAPE = -2,    /* shape not supported by the kernels (see each function) */
    STDIFF_ERR_ALIGN = -3,    /* pointer or leading dimension not 16-byte aligned */
    STDIFF_ERR_STATE = -4,    /* handle misuse (graph not captured, ...) */
    STDIFF_ERR_SCRATCH = -5   /* problem larger than the library's static scratch buffer (stdiff_gemm_static_scratch) */
};

/* GEMM epilogues: v = sum_k A[m,k] W[n,k] + bias[n] */
enum {
    STDIFF_EPI_BIAS = 0,      /* C = v                                   (nn.Linear) */
    STDIFF_EPI_SILU = 1,      /* C = silu(v)                             (time MLP, stDiff_model.py:53-57) */
    STDIFF_EPI_GELU = 2,      /* C = gelu_erf(v)                         (timm Mlp fc1+act, stDiff_model.py:100-103) */
    STDIFF_EPI_GATE_RES = 3,  /* C = res[m,n] + gate[m,n] * v  (fp32 C)  (gated residual, stDiff_model.py:114,116) */
    STDIFF_EPI_ADD = 4,       /* C = v + res[m,n]                        (c = t + y, stDiff_model.py:237) */
    STDIFF_EPI_GELU_SAVE = 5, /* C = gelu_erf(v), C2 = v (C2 required)   (training: keep the pre-activation) */
    STDIFF_EPI_ACCUM = 6,     /* C = C + v  (fp32 C)                      (gradient accumulation) */
    /* adaLN projection fused with the LayerNorm + modulate it feeds (stDiff_model.py:105-108 with :41-44,96,99,126):
     * the N = 2*Hn weight rows come in 256-row tiles [128 shift rows | the 128 scale rows of the same features], and
     *   C[m, 128*j + c] = LN(x[m,:])[128*j + c] * (1 + scale) + shift,   j = tile, c < 128,   C: [M, N/2] 16-bit,
     * with x = `res` (fp32, the residual stream), LN from the per-row statistics `ln_parts` (below) and shift / scale
     * rounded to the 16-bit format first: what stdiff_gemm(EPI_BIAS) into a 16-bit buffer followed by
     * stdiff_ln_modulate gives (bit for bit when the statistics come from stdiff_row_stats), without writing shift /
     * scale or re-reading them.  16-bit operands, N % 256 == 0, 16-byte aligned rows. */
    STDIFF_EPI_LN_MOD = 7,
    /* backward of GELU fused into the GEMM that produces its upstream gradient (dX of mlp.fc2, training):
     *   C = round(v) * gelu_erf'(u_pre[m,n]),  u_pre given through `gate` / `ldgate` in the operand dtype,
     * v rounded to the output dtype first -- bit for bit what stdiff_gemm(EPI_BIAS) followed by stdiff_gelu_bwd gives. */
    STDIFF_EPI_GELU_BWD = 8
};

int stdiff_abi_version(void);
/* 16-bit operand format of this build: 0 = bf16, 1 = IEEE fp16 (no counterpart in the reference, which is fp32). */
int stdiff_h16_format(void);
const char* stdiff_error_string(int status);

/* ---------------------------------------------------------------- dense contractions (MFMA)
 * C[M,N] = epi(A[M,K] . W[N,K]^T + bias): the nn.Linear layers of stDiff_model.py
 * (:18,20 qkv/proj; :54,56 time MLP; :107,132 adaLN; :128 out; :177,180 in/cond; timm Mlp fc1/fc2).
 * A and W share a_dtype; K % 64 == 0 (bf16) or K % 16 == 0 (f32); lda/ldw multiples of 8 (bf16) / 4 (f32)
 * elements and A/W 16-byte aligned; M, N arbitrary. gate has a_dtype, res is fp32.  With ANY epilogue a non-NULL
 * C2 (a_dtype, ldc2) additionally receives v itself -- the training forward keeps pre-activations / branch outputs. */
enum { STDIFF_ROW_MAJOR = 0, STDIFF_K_MAJOR = 1 };
/* A GEMM result left as split-K partial products (fp32):  v[m][n] = sum_{s < slices} ws[s * slice_stride + m * ldws + n].
 * Filled by stdiff_gemm through stdiff_gemm_args.defer; consumed by stdiff_ln_modulate_split,
 * stdiff_ln_modulate_bwd_split and stdiff_attention_bwd_split, which sum the slices in slice order (deterministic, the
 * order of the reduce launch they replace).  No counterpart in the reference: it removes a launch from the dependent
 * chain of a 512-cell training iteration. */
typedef struct { const float* ws; int ldws; int slices; long long slice_stride; } stdiff_split_result;
typedef struct {
    int M, N, K;
    const void* A; int lda; int a_dtype;
    const void* W; int ldw;
    const float* bias;
    void* C; int ldc; int c_dtype;
    int epi;
    const void* gate; int ldgate;
    const float* res; int ldres;
    void* C2; int ldc2;
    /* optional split-K scratch (fp32, 16-byte aligned, caller-owned; NULL = never split).  Outputs with few tiles and a
     * long K (the batch-sized GEMMs of a training step) are cut along K into partial products there and summed in a
     * fixed order by a second kernel that applies the epilogue: deterministic, no atomics. */
    void* workspace; long long workspace_bytes;
    /* operand layouts (0 = STDIFF_ROW_MAJOR: [M][K] / [N][K] as above; 1 = STDIFF_K_MAJOR: the operand is [K][M] resp.
     * [K][N] with lda / ldw the stride of a k-row).  The backward contractions dW = dY^T X (both K-major: K = the
     * batch) and dX = dY W (W K-major) run without transposed copies.  k_valid (0 = K): rows k >= k_valid of K-major
     * operands count as zero, so a batch that is not a multiple of the K-slab needs no padded copy either; row-major
     * operands must then really have K columns. */
    int a_layout, w_layout, k_valid;
    /* Row statistics of a residual stream [M, Hn] as partial moments, fp32 [M][parts][2] = (mean_p, M2_p) of `parts`
     * equal column ranges (M2 = sum of squared deviations from mean_p); mean = avg(mean_p), biased variance =
     * (sum M2_p + cols_per_part * sum (mean_p - mean)^2) / Hn.
     *   ln_parts / ln_nparts: read by STDIFF_EPI_LN_MOD for its `res` (Hn = N/2); 1 <= ln_nparts <= 8, Hn % ln_nparts == 0.
     *   stats_out: if non-NULL (fp32 C, N % 256 == 0, 16-bit operands) this GEMM also WRITES the partial moments of the
     *   rows of C it produces, one part per 256-column tile ([M][N/256][2]): the epilogue that adds the gated branch
     *   to the residual stream hands the next LayerNorm its statistics, and no pass over C is needed for them. */
    const float* ln_parts; int ln_nparts;
    float* stats_out;
    /* defer (host pointer, may be NULL): when the launch splits its K axis (few tiles, long K, a workspace) and epi is
     * STDIFF_EPI_BIAS or STDIFF_EPI_GATE_RES, the reduce launch is SKIPPED, the partial products stay in the workspace
     * and *defer describes them: the consumer sums the slices and applies the epilogue itself (the *_split entry
     * points below).  defer->slices == 0 on return: the launch completed normally (C is written). */
    stdiff_split_result* defer;
} stdiff_gemm_args;
int stdiff_gemm(const stdiff_gemm_args* a, void* stream);

/* stdiff_gemm WITHOUT ANY ALLOCATION.  No counterpart in the reference: this entry point exists for callers that cannot
 * allocate device memory -- code under hipGraph stream capture, tight serving loops, harnesses that forbid hipMalloc --
 * and therefore cannot supply the split-K `workspace` either.  Same arguments, same kernels and the same results as
 * stdiff_gemm called with a workspace of STDIFF_STATIC_SCRATCH_BYTES (a->workspace / a->workspace_bytes are ignored):
 * the fp32 partial products of a K-split go to a file-scope __device__ array inside the library (64 MiB per GPU the
 * library runs on, held until the process ends), whose per-device address is looked up once with
 * hipGetSymbolAddress on the first call on that device (make that first call outside stream capture).
 * Limit: a shape whose K axis the host would split (few 128 x 128 tiles, long K) needs at least 3 slices of
 * M * round_up(N, 4) * 4 bytes in the buffer, else STDIFF_ERR_SCRATCH and nothing is launched; shapes that are not split
 * (every GEMM with more tiles than a third of the CUs, e.g. the sampler's 131072-row launches) use no scratch and have
 * no size limit here.
 * All calls on one GPU share that GPU's array: calls that overlap on one GPU (two streams, two threads) race; calls
 * on one stream are ordered.  Nothing in the array needs to start at zero (every slice is written before it is read). */
#define STDIFF_STATIC_SCRATCH_BYTES (64u << 20)
int stdiff_gemm_static_scratch(const stdiff_gemm_args* a, void* stream);

/* The STDIFF_GEMM_TILE / _CUS / _SPLIT_TAIL and STDIFF_SPLITK_MAX / _BLOCKS / _MIN knobs (A/B timing, tests) are read
 * from the environment once per process, at the first stdiff_gemm; this call reads them again.  No counterpart in the
 * reference (a debugging hook).  Safe against concurrent launches (the record is swapped atomically). */
void stdiff_reload_env(void);

/* out[c, r] = in[r, c] for r < rows, and 0 for rows <= r < rows_pad (same dtype, ld_out >= rows_pad): turns the
 * TN / NN contractions of the backward pass (dW = dY^T X, dX = dY W) into NT GEMMs with a zero-padded K. */
int stdiff_transpose(const void* in, int ld_in, void* out, int ld_out, int rows, int cols, int rows_pad, int dtype,
                     void* stream);

/* ---------------------------------------------------------------- cell-axis attention
 * Attention2.forward core (stDiff_model.py:25-34): per chunk (consecutive rows cu_rows[c]..cu_rows[c+1]) and
 * head, softmax(q k^T d^-0.5) v over the CELLS of that chunk; never materialises [h,C,C].
 * qkv: [R, 3H] (ld = ldqkv) in the nn.Linear column order [q|k|v], each heads x d; out: [R, H].
 * d = H/heads must be 32, 64 or 128.  lse (optional, fp32 [R, heads]) receives log2-sum-exp rows for backward. */
int stdiff_attention_fwd(const void* qkv, int ldqkv, void* out, int ldo, float* lse, const int32_t* cu_rows,
                         int n_chunks, int max_chunk_rows, int heads, int head_dim, int dtype, void* stream);

/* backward of the above: dqkv[R,3H] (all three column blocks written) from dout[R,H], recomputing P from qkv and
 * lse in two passes (dQ with the query on the lane; dK/dV with the key on the lane; no atomics).
 * delta (fp32 [R, heads]) is scratch supplied by the caller; R = total_rows = cu_rows[n_chunks]. */
int stdiff_attention_bwd(const void* qkv, int ldqkv, const void* out, int ldo, const void* dout, int lddo,
                         const float* lse, float* delta, void* dqkv, int lddqkv, const int32_t* cu_rows, int n_chunks,
                         int max_chunk_rows, int total_rows, int heads, int head_dim, int dtype, void* stream);

/* ---------------------------------------------------------------- bandwidth-bound fused kernels
 * y = LayerNorm(x; eps 1e-6, no affine, biased var) * (1 + scale) + shift   (modulate, stDiff_model.py:41-44,96,99,126)
 * x fp32 [R,H]; shift/scale rows of the adaLN output (dtype mod_dtype, leading dim ldmod); y [R,H] of out_dtype.
 * Optionally writes mean/rstd (fp32 [R]) for backward. */
int stdiff_ln_modulate(const float* x, int ldx, const void* shift, const void* scale, int ldmod, int mod_dtype,
                       void* y, int ldy, int out_dtype, float* mean, float* rstd, int R, int H, void* stream);

/* The statistics pass of the above alone, in the partial-moment form of stdiff_gemm_args with ONE part:
 * parts[r] = (mean, sum (x - mean)^2) of x[r, :H] (same reduction order as stdiff_ln_modulate, so rsqrt(M2/H + 1e-6) is
 * its rstd bit for bit); feeds the STDIFF_EPI_LN_MOD epilogue of stdiff_gemm. */
int stdiff_row_stats(const float* x, int ldx, float* parts, int R, int H, void* stream);

/* s[r,:] = silu(c[r,:])  with  c[r,:] = temb[*t_index, :] + ycond[r,:]  when temb != NULL (sampling: every row of a
 * reverse step shares t, model/sample.py:11), else c = ycond (training: c precomputed).   (nn.SiLU of
 * adaLN_modulation, stDiff_model.py:105-108,130-133; c = t + y, :237) */
int stdiff_silu_cond(const float* temb, const int32_t* t_index, const float* ycond, int ldy, void* s, int lds,
                     int out_dtype, int R, int H, void* stream);

/* sinusoidal timestep embedding (stDiff_model.py:61-80): t int64 [R] -> [R,256] of out_dtype, [cos(t f) | sin(t f)].
 * freqs: device fp32 [128] = exp(-ln(1e4) k/128), computed by the caller with the reference's fp32 op sequence. */
int stdiff_timestep_embedding(const int64_t* t, const float* freqs, void* out, int ldo, int out_dtype, int R,
                              void* stream);

/* dst[R, ldd] (dtype) = src[R, C] fp32, columns C..Cpad-1 zero-filled (operand packing for the MFMA kernels). */
int stdiff_pack_rows(const float* src, int lds, void* dst, int ldd, int dtype, int R, int C, int Cpad, void* stream);

/* Shuffled mini-batch from device-resident data (the DataLoader of process/data.py:64-75 with shuffle=True, minus the
 * host: TensorDataset.__getitem__ + default collate = a row gather): da[r,:] = a[idx[r],:] and, when b != NULL,
 * db[r,:] = b[idx[r],:] for r < R.  fp32, idx int64 on the device. */
int stdiff_gather_rows2(const float* a, int lda, const float* b, int ldb, const int64_t* idx, float* da, int ldda,
                        float* db, int lddb, int R, int C, void* stream);

/* Expression pre-processing per cell (test-stDiff.py:47-55: sc.pp.normalize_total(target_sum) -> sc.pp.log1p ->
 * scale() = MaxAbsScaler over the cell's genes, process/data.py:79-83), with data_augment's fixed offset
 * (process/data.py:86-100) as add_const:  y[r,:] = log1p((x[r,:] + add_const) * target_sum / rowsum) / rowmax.
 * Zero-sum / zero-max rows divide by 1.  fp32, y may alias x.
 * target_sum == 0: scale() alone (no normalize_total / log1p, add_const ignored): y[r,:] = x[r,:] / max|x[r,:]|, a true
 * fp32 division as MaxAbsScaler performs it (bit-equal to the reference on fp32 input, golden G10). */
int stdiff_preprocess_rows(const float* x, int ldx, float* y, int ldy, int R, int G, float add_const, float target_sum,
                           void* stream);

/* ---------------------------------------------------------------- diffusion arithmetic
 * One reverse step + known-gene splice, fused (NoiseScheduler.step, stDiff_scheduler.py:103-139 ->
 * reconstruct_x0 :68-78, q_posterior :80-90, get_variance :92-100; splice model/sample.py:94-95):
 *   x0 = clamp(a*x - b*eps, -1, 1)   (pred_type 0 'noise')   |   x0 = eps   (pred_type 1 'x_start')
 *   x  = (c1*x0 + c2*x + sigma*z) * (1-mask) + mask*gt       with (a,b,c1,c2,sigma) = coef[t*5 .. t*5+4]
 * t = *t_dev (device int32, so a captured graph can be replayed) ; z = z_base + z_index*z_rows*G when z_base != NULL
 * (z_index = t_first - t; z_rows = rows of one step's noise matrix, 0 = R: a call that updates a row slab of a larger
 * sample passes the slab's first row of step 0 as z_base and the whole sample's row count as z_rows), otherwise
 * Philox4x32-10 + Box-Muller keyed by (seed, t, global element index).
 * Also refreshes xp (the [R,Gp] MFMA operand copy of x) when xp != NULL. Arithmetic order equals the reference's
 * (no FMA contraction) so fp32 results are bit-identical to it. */
typedef struct {
    float* x; const float* eps; const float* gt; const float* mask; const float* coef;
    const int32_t* t_dev; int t_first;
    const float* z_base; uint64_t seed; int64_t row_offset;   /* row_offset: global index of row 0 (sharded sampling) */
    void* xp; int ldxp; int xp_dtype;
    int R, G, pred_type;
    int64_t z_rows;
} stdiff_step_args;
int stdiff_ddpm_step(const stdiff_step_args* a, void* stream);
/* *t_dev -= 1 (one thread; closes a captured reverse step). */
int stdiff_advance_timestep(int32_t* t_dev, void* stream);

/* `blocks` one-wave workgroups that each wait `microseconds` (<= 100 ms) on the realtime counter and exit.  No
 * counterpart in the reference: the sampler runs two chunk groups on two HIP streams, and two such kernels launched
 * on two streams finish in about one period only if the streams' hardware queues really dispatch concurrently --
 * the host uses that to choose its lane streams. */
int stdiff_spin_probe(int blocks, int microseconds, void* stream);

/* x_T init of sample_stDiff (model/sample.py:56-60): x = z*(1-mask) + gt*mask, z injected or Philox. */
int stdiff_init_sample(float* x, const float* z, uint64_t seed, int64_t row_offset, const float* gt, const float* mask,
                       void* xp, int ldxp, int xp_dtype, int R, int G, void* stream);

/* forward diffusion + train splice (stDiff_scheduler.py:141-148; stDiff_train.py:65-70):
 * x_noisy = (sa[t_r]*x + sb[t_r]*noise)*(1-mask) + x*mask, written as fp32 and/or as the packed operand xp. */
int stdiff_add_noise_splice(const float* x, const float* noise, const int64_t* t, const float* sqrt_ac,
                            const float* sqrt_1mac, const float* mask, float* x_noisy, void* xp, int ldxp, int xp_dtype,
                            int R, int G, void* stream);

/* masked MSE (stDiff_train.py:75): loss = mean_{R*G}((noise*(1-m) - pred*(1-m))^2) accumulated into *loss (caller
 * zeroes it) and dpred = 2*(1-m)^2*(pred-noise)/(R*G) * grad_scale, as fp32 and/or the packed operand dp. */
int stdiff_masked_mse(const float* pred, const float* noise, const float* mask, float* loss, float* dpred, void* dp,
                      int lddp, int dp_dtype, float grad_scale, int R, int G, void* stream);

/* ---------------------------------------------------------------- backward helpers (training)
 * d(LN-modulate): given dy [R,H] (dtype), x, mean, rstd, scale, shift-layout as in forward:
 *   dx[R,H] (fp32, accumulated into dx when accumulate != 0), dshift = dy, dscale = dy * xhat  (dtype, into dmod cols). */
int stdiff_ln_modulate_bwd(const void* dy, int lddy, int dtype, const float* x, int ldx, const float* mean,
                           const float* rstd, const void* scale, int ldmod, float* dx, int lddx, int accumulate,
                           void* dshift, void* dscale, int lddmod, int R, int H, void* stream);
/* The same followed by the gated-residual backward that comes next in the chain (x_next = x + gate * branch, DiTblock,
 * stDiff_model.py:114,116; d x_next is the dx formed here): dbranch = dx * gate, dgate = dx * branch, in one launch --
 * bit for bit what stdiff_ln_modulate_bwd + stdiff_gate_bwd produce. */
int stdiff_ln_modulate_bwd_gate(const void* dy, int lddy, int dtype, const float* x, int ldx, const float* mean,
                                const float* rstd, const void* scale, int ldmod, float* dx, int lddx, int accumulate,
                                void* dshift, void* dscale, int lddmod, const void* gate, int ldgate, const void* branch,
                                int ldbr, void* dbranch, int lddb, void* dgate, int lddg, int R, int H, void* stream);
/* The three consumers of a deferred split-K result (stdiff_split_result).  Results equal the unfused sequence
 * (stdiff_gemm with its reduce launch, then the plain entry point) bit for bit.
 *  - forward:  v = sum + bias;  x = res + gate * v (gate == NULL: x = v);  v_out (mod_dtype, optional) = v;  then
 *    stdiff_ln_modulate of x.  (attn.proj / mlp.fc2 with their gated residual, followed by the next LayerNorm.)
 *  - backward: dy = the sum rounded to `dtype`, then stdiff_ln_modulate_bwd_gate (gate == NULL: stdiff_ln_modulate_bwd).
 *  - attention backward: dout = the sum rounded to `dtype` (written to `dout`, [R, >= heads*d]), then stdiff_attention_bwd. */
int stdiff_ln_modulate_split(const stdiff_split_result* sp, const float* bias, const void* gate, int ldgate,
                             const float* res, int ldres, float* x, int ldx, void* v_out, int ldv, const void* shift,
                             const void* scale, int ldmod, int mod_dtype, void* y, int ldy, int out_dtype, float* mean,
                             float* rstd, int R, int H, void* stream);
int stdiff_ln_modulate_bwd_split(const stdiff_split_result* sp, int dtype, const float* x, int ldx, const float* mean,
                                 const float* rstd, const void* scale, int ldmod, float* dx, int lddx, int accumulate,
                                 void* dshift, void* dscale, int lddmod, const void* gate, int ldgate, const void* branch,
                                 int ldbr, void* dbranch, int lddb, void* dgate, int lddg, int R, int H, void* stream);
int stdiff_attention_bwd_split(const void* qkv, int ldqkv, const void* out, int ldo, const stdiff_split_result* sp,
                               void* dout, int lddo, const float* lse, float* delta, void* dqkv, int lddqkv,
                               const int32_t* cu_rows, int n_chunks, int max_chunk_rows, int total_rows, int heads,
                               int head_dim, int dtype, void* stream);
/* gated residual backward: dbranch = dh*gate (dtype), dgate = dh*branch (dtype) ; dh fp32 */
int stdiff_gate_bwd(const float* dh, int lddh, const void* gate, int ldgate, const void* branch, int ldbr,
                    void* dbranch, int lddb, void* dgate, int lddg, int dtype, int R, int H, void* stream);
/* du = dg * gelu'(u_pre)  (erf GELU) */
int stdiff_gelu_bwd(const void* dg, const void* u_pre, void* du, int ld, int dtype, int R, int C, void* stream);
/* dc (+)= ds * silu'(c)   (nn.SiLU backward: conditioning vector stDiff_model.py:106,131 and time MLP :55);
 * each of ds / c / dc may be fp32 or bf16. */
int stdiff_silu_bwd(const void* ds, int ldds, int ds_dtype, const void* c, int ldc, int c_dtype, void* dc, int lddc,
                    int dc_dtype, int accumulate, int R, int H, void* stream);
/* db[n] (+)= sum_m dY[m,n]   (bias gradients) */
int stdiff_colsum(const void* dY, int ld, int dtype, float* db, int accumulate, int R, int C, void* stream);

/* ---------------------------------------------------------------- optimizer (stDiff_train.py:47,78-80)
 * All trainable parameters live in ONE flat fp32 buffer (so do grads and the two Adam moments); under data
 * parallelism the flat gradient buffer is what RCCL all-reduces. */
/* *sumsq += sum g^2 (caller zeroes it). */
int stdiff_grad_sumsq(const float* grads, int64_t n, float* sumsq, void* stream);
/* nn.utils.clip_grad_norm_(params, max_norm) + torch.optim.AdamW step in one pass (28 B/param):
 *   step = ++*step_dev;  norm = sqrt(*sumsq) * grad_scale;  coef = min(1, max_norm / (norm + 1e-6)) * grad_scale
 *   g' = g*coef;  p *= 1 - lr*wd;  m = b1 m + (1-b1) g';  v = b2 v + (1-b2) g'^2;
 *   p -= lr/(1-b1^step) * m / (sqrt(v)/sqrt(1-b2^step) + eps)
 * grad_scale = 1/world_size when `grads` holds an all-reduced SUM.  The step counter and the derived scalars live on
 * the device (step_dev int32[1]; scal fp32[4] = {coef, lr/bc1, sqrt(bc2), norm}) so a captured train step replays. */
typedef struct {
    float* params; const float* grads; float* exp_avg; float* exp_avg_sq; int64_t n;
    const float* sumsq; int32_t* step_dev; float* scal;
    float max_norm, grad_scale, lr, beta1, beta2, eps, weight_decay;
    void* bf16_mirror;   /* optional bf16 [n]: receives the updated parameters (the next forward's packed operands) */
} stdiff_adamw_args;
int stdiff_clip_adamw(const stdiff_adamw_args* a, void* stream);
/* The same update in two parts: the scalars once (step counter, norm, clip factor, bias corrections), then any number
 * of element ranges [lo, lo + count) of the flat buffers (lo % 4 == 0), each with the arithmetic above.  A caller that
 * issues one range per layer on its own stream lets the NEXT iteration's forward start on a layer as soon as that
 * layer's parameters are updated (stdiff_train_acts.param_ready).  stdiff_clip_adamw = prepare + range(0, n). */
int stdiff_adamw_prepare(const stdiff_adamw_args* a, void* stream);
int stdiff_adamw_range(const stdiff_adamw_args* a, int64_t lo, int64_t count, void* stream);

/* ---------------------------------------------------------------- per-gene accuracy statistics (test-stDiff.py:203-364)
 * One pass over imputed x and measured y ([N, G] fp32): partials[chunk][k][g], k = sum x, sum y, sum xy, sum xx,
 * sum yy, max x, max y in float64 for the rows of slab `chunk` (rows cut into `chunks` equal slabs; the caller adds
 * the slabs in order: deterministic).  Pearson (np.corrcoef / scipy pearsonr), cosine, RMSE after zscore
 * (CalculateMeteics.RMSE :345-364) and SSIM after scale_max (cal_ssim :203-221, SSIM :271-298) are closed forms of
 * these.  partials: device float64 [chunks, 7, G]. */
int stdiff_column_moments(const float* x, int ldx, const float* y, int ldy, double* partials, int chunks, int N, int G,
                          void* stream);
/* Jensen-Shannon divergence of the sum-normalised columns (CalculateMeteics.JS :318-343 with scale_plus :237-242 and
 * scipy.stats.entropy): partials[chunk][g] in float64; inv_sum_* = 1 / column sums (device float64 [G]). */
int stdiff_column_js(const float* x, int ldx, const float* y, int ldy, const double* inv_sum_x, const double* inv_sum_y,
                     double* partials, int chunks, int N, int G, void* stream);

/* Spearman per gene (CalculateMeteics.SPCC :300-316, scipy.stats.spearmanr): xt / yt are the TRANSPOSED matrices
 * ([G][ldt] fp32, gene g contiguous at g * ldt; stdiff_transpose makes them).  partials[g][b][3] (float64) = sums over
 * the elements of block b (1024 consecutive cells) of rx'^2, ry'^2, rx' ry', rx' = average rank - (N+1)/2 (exact
 * half-integers, ranks by counting).  blocks must be ceil(N / STDIFF_RANK_CELLS_PER_BLOCK); G <= 65535; cost grows as
 * N^2 G.  rho_g = S_xy / sqrt(S_xx S_yy) with the blocks added in order. */
#define STDIFF_RANK_CELLS_PER_BLOCK 1024
int stdiff_rank_corr(const float* xt, const float* yt, long long ldt, double* partials, int blocks, int N, int G,
                     void* stream);

/* ---------------------------------------------------------------- whole-denoiser forward (inference)
 * DiT_stDiff.forward after the conditioning (stDiff_model.py:240-243): in_layer, depth x DiTblock, FinalLayer,
 * on R rows split into chunks (attention is per chunk).  Inputs: xp = packed x [R,Gp], s = silu(c) [R,H].
 * All buffers are caller-owned (torch).  Enqueues ~8 kernels per block on `stream`.
 * Two equivalent forms (bit-identical results): the plain one (adaLN GEMM -> [R,6H]; LayerNorm+modulate kernels), and the
 * fused-LN one for 16-bit operands, where each LayerNorm+modulate is the EPILOGUE of the adaLN GEMM that produces its
 * shift / scale (STDIFF_EPI_LN_MOD): the [R,6H] modulation matrix is never written, and the LayerNorm's statistics come
 * from the epilogue that wrote the residual stream (stats_out; H % 256 == 0) or from stdiff_row_stats.  The two forms
 * agree to rounding (bit for bit when H % 256 != 0); a given build / shape always takes the same one. */
typedef struct {
    const void* qkv_w; const float* qkv_b;
    const void* proj_w; const float* proj_b;
    const void* fc1_w; const float* fc1_b;
    const void* fc2_w; const float* fc2_b;
    const void* ada_w; const float* ada_b;
    /* fused-LN form of the adaLN projection (optional: NULL = not packed).  ln1 / ln2: [2H, H] rows in 256-row tiles
     * [128 shift | 128 scale] of (shift_msa, scale_msa) resp. (shift_mlp, scale_mlp), biases permuted alike;
     * gate: [2H, H] = [gate_msa ; gate_mlp] rows. */
    const void* ln1_w; const float* ln1_b;
    const void* ln2_w; const float* ln2_b;
    const void* gate_w; const float* gate_b;
} stdiff_block_weights;
typedef struct {
    int dtype, G, Gp, H, heads, depth;
    const void* in_w; const float* in_b;       /* [H,Gp] */
    const void* out_w; const float* out_b;     /* [G,H]  */
    const void* fada_w; const float* fada_b;   /* [2H,H] */
    const stdiff_block_weights* blocks;        /* host array [depth] */
    const void* lnf_w; const float* lnf_b;     /* final adaLN in the fused-LN tile order (optional) */
} stdiff_dit_weights;
typedef struct {
    int R, n_chunks, max_chunk_rows;
    const int32_t* cu_rows;                    /* device int32 [n_chunks+1] */
    float* h;                                  /* [R,H]  residual stream */
    void* mod; void* xn; void* qkv; void* o; void* u; void* fmod;   /* [R,6H] [R,H] [R,3H] [R,H] [R,4H] [R,2H] */
    /* fused-LN path (taken when both are non-NULL and the weights carry their fused form; mod / fmod may then be
     * NULL): partial row moments fp32 [R][max(1, H/256)][2] and the two gates of a block [R, 2H] */
    float* ln_parts; void* gates;
} stdiff_dit_workspace;
int stdiff_dit_forward(const stdiff_dit_weights* w, const stdiff_dit_workspace* ws, const void* xp, const void* s,
                       float* eps, int ldeps, void* stream);

/* ---------------------------------------------------------------- grouped weight gradients
 * The weight and bias gradients of up to 8 linear layers over ONE batch of C cells in one launch (what
 * loss.backward() leaves in .grad of each nn.Linear, stDiff_train.py:77):  gw_i[M_i, N_i] = dY_i[:, :M_i]^T X_i[:, :N_i],
 * gb_i[M_i] = column sums of dY_i (NULL: not wanted).  dY_i [C, >= M_i] and X_i [C, >= N_i] are read as they lie
 * (row = cell); gradients are fp32 and OVERWRITTEN.  The K axis (cells) is never split: results do not depend on the
 * number of CUs.  M_i % 8 (fp32: % 4), N_i % 8, 16-byte aligned rows; else STDIFF_ERR_SHAPE / STDIFF_ERR_ALIGN. */
#define STDIFF_DW_GROUP_MAX 8
typedef struct {
    int M, N;
    const void* dY; int lddy;
    const void* X; int ldx;
    float* gw; int ldgw;
    float* gb;
} stdiff_dw_problem;
typedef struct {
    int n, C, dtype;                    /* problems, cells, operand dtype (STDIFF_F32 / STDIFF_BF16) */
    stdiff_dw_problem p[STDIFF_DW_GROUP_MAX];
    float* sq_parts;                    /* NULL, or stdiff_dw_group_parts(g) floats, ALL overwritten with plain stores: one
                                           per tile, the sum of squares of what the tile writes (weights and biases;
                                           padding slots 0).  Their sum is the launch's share of clip_grad_norm_'s squared
                                           norm (stDiff_train.py:78), taken from the accumulators instead of a pass over
                                           the gradients; add them up with stdiff_sum_parts */
} stdiff_dw_group_args;
int stdiff_dw_group(const stdiff_dw_group_args* g, void* stream);
long long stdiff_dw_group_parts(const stdiff_dw_group_args* g);       /* floats of g->sq_parts; < 0: bad arguments */
/* *out = (accumulate ? *out : 0) + parts[0] + ... + parts[n - 1], one workgroup, a fixed summation order (the same bits on
 * every run).  n >= 0; no reference counterpart (it finishes clip_grad_norm_'s norm, stDiff_train.py:78). */
int stdiff_sum_parts(const float* parts, long long n, float* out, int accumulate, void* stream);

/* ---------------------------------------------------------------- whole-denoiser training pass
 * The forward of stDiff_train.py:72 (DiT_stDiff.forward, stDiff_model.py:231-243, on one mini-batch = one attention
 * chunk) keeping what the backward needs, and the backward of :77 (loss.backward() through the denoiser), each as ONE
 * enqueue of ~110 resp. ~190 launches.  All buffers are caller-owned.  16-bit activations have w->dtype; "fp32" ones
 * are float.  Shapes: C = cells of the batch, H hidden, G genes, Gp = G rounded up to 64 (operand padding). */
typedef struct {
    const void* w;          /* packed operand [out, in_padded] of w->dtype (nn.Linear layout) */
    const float* b;         /* bias fp32 [out] */
    float* gw; float* gb;   /* gradients fp32 [out, in] (UNPADDED, contiguous) and [out]: written by stdiff_dit_backward */
} stdiff_train_linear;
typedef struct { stdiff_train_linear qkv, proj, fc1, fc2, ada; } stdiff_train_block;   /* [3H,H] [H,H] [4H,H] [H,4H] [6H,H] */
typedef struct {
    int dtype, G, Gp, H, heads, depth;
    stdiff_train_linear t0, t2;                 /* time_emb.mlp.0 [H,256], time_emb.mlp.2 [H,H] */
    stdiff_train_linear cond, in;               /* cond_layer.0, in_layer.0: [H,Gp] packed, gradients [H,G] */
    stdiff_train_linear fada, out;              /* out_layer.adaLN_modulation.1 [2H,H], out_layer.linear [G,H] */
    const stdiff_train_block* blocks;           /* host array [depth] */
    const float* freqs;                         /* as for stdiff_timestep_embedding */
} stdiff_train_weights;
struct stdiff_grad_ranges { float* sumsq; const float* lo[64]; long long count[64]; int in_group; float* parts; long long parts_count; };
typedef struct {            /* saved by block k's forward */
    float* h; float* hmid;                      /* fp32 [C,H]: block input, after the attention branch */
    void* mod;                                  /* [C,6H] adaLN output */
    void* xn1; void* xn2; void* qkv; void* o;   /* [C,H] [C,H] [C,3H] [C,H] */
    float* lse;                                 /* fp32 [C,heads] */
    void* pa; void* pm;                         /* [C,H] branch outputs before the gate (proj / fc2) */
    void* upre; void* u;                        /* [C,4H] fc1 before / after GELU */
    float* mean1; float* rstd1; float* mean2; float* rstd2;   /* fp32 [C] */
} stdiff_train_block_acts;
typedef struct {
    int C;
    const int32_t* cu_rows;                     /* device int32 {0, C} */
    void* xp; void* yp;                         /* packed inputs [C,Gp]: x_noisy and x_cond */
    void* e0; void* a1; void* e1;               /* [C,256] [C,H] [C,H]: time MLP */
    float* temb; float* c; void* s;             /* fp32 [C,H], fp32 [C,H], [C,H] = silu(c) */
    const stdiff_train_block_acts* blocks;      /* host array [depth] */
    float* h_last;                              /* fp32 [C,H]: output of the last block */
    void* fmod; void* xnf; float* meanf; float* rstdf;     /* [C,2H] [C,H] fp32 [C] fp32 [C] */
    float* eps;                                 /* fp32 [C,G]: the forward's result */
    /* backward: deps = d loss / d eps as the packed operand [C,Gp] (stdiff_masked_mse writes it); the rest is scratch */
    void* deps; float* dh; void* dxn; void* dmod[2]; void* dfmod; void* dbr; void* du; void* dupre; void* do_; void* dqkv;
    float* delta; float* ds32; float* dc; void* dcT; void* de1; void* da1; void* dhT;
    /* [C,Gp] [C,H]f32 [C,H] 2x[C,6H] [C,2H] [C,H] [C,4H] [C,4H] [C,H] [C,3H] f32[C,heads] f32[C,H] f32[C,H] [C,H] x4 */
    void* tr[2][2];        /* G % 8 != 0 only: [stream: main, side][dY^T, X^T] transposed copies [max(Gp,H), round_up(C,64)] */
    void* splitk_ws; long long splitk_bytes;             /* split-K scratch of the launches on `stream` (may be NULL) */
    void* splitk_ws_side; long long splitk_side_bytes;   /* ... of the launches on the side stream */
    /* Optional, all four or none: three more [C,H] and one more [C,3H] scratch.  With them the backward issues ONE
     * stdiff_dw_group launch per DiT block (its five weight and bias gradients) instead of five jobs of three launches:
     * the dY operands of a block (dbr of either branch, dupre with `du` as its twin, dqkv, dmod) then alternate between
     * two sets so that a block's group can run while the chain is already in the next block. */
    void* dbr_b[3]; void* dqkv_b;
    /* != 0: never leave a GEMM result as split-K partial products for its consumer (stdiff_gemm_args.defer): every split
     * GEMM is followed by its reduce launch, as before ABI 11 (A/B timing; results are bit-identical either way) */
    int no_defer;
    /* Optional host array of 4 * depth + 2 hipEvent_t (NULL: none; NULL entries are skipped): stdiff_dit_train_forward
     * makes its streams wait for param_ready[0] before the first use of the embedding layers' parameters (time MLP, cond,
     * in); for block k for param_ready[1 + 4k] before its adaLN projection (and with it before the block), and -- for a
     * block whose parameters are updated in pieces, else NULL -- for [2 + 4k] before qkv (proj is in the same piece),
     * [3 + 4k] before fc1, [4 + 4k] before fc2; and for param_ready[4 * depth + 1] before the final layer's: the events an
     * optimizer records after updating those ranges on another stream (stdiff_adamw_range), so that the update of
     * iteration i - 1 runs under the forward of iteration i.  Already-recorded events only.  Every wait is a
     * cross-stream edge that costs several microseconds: pieces pay only where the update, not the forward, sets the pace. */
    void* const* param_ready;
    /* Optional (NULL: none; honoured by the grouped backward only): the squared gradient norm accumulated range by range
     * as the backward finishes each gradient bucket -- *sumsq += sum g^2 over [lo[i], lo[i] + count[i]) for bucket i of
     * stdiff_bucket_fn's numbering (depth + 2 buckets), issued on the stream that ran the bucket's last launches, so the
     * norm pass of clip_grad_norm_ (stDiff_train.py:78) hides under the input-gradient chain.  The caller zeroes *sumsq
     * before the backward and must not use this under data parallelism (the norm is that of the REDUCED gradient).
     * in_group != 0 (and G % 8 == 0): no range pass at all -- every stdiff_dw_group launch of the backward leaves the
     * squares of what it writes in its slice of `parts` (stdiff_dw_group_args.sq_parts; parts_count >=
     * stdiff_dit_backward_parts(w) floats, else STDIFF_ERR_ARG) and a stdiff_sum_parts launch behind each group, on the
     * group's stream, adds them to *sumsq in a fixed order; lo / count are then unused. */
    const struct stdiff_grad_ranges* grad_ranges;
} stdiff_train_acts;
/* side_stream (NULL = none): the adaLN projections of the forward and the weight-gradient jobs of the backward are
 * issued there and joined with events the library creates once per device (hipEventDisableTiming; no device memory):
 * under stream capture they become parallel branches of the graph.  Results do not depend on it. */
int stdiff_dit_train_forward(const stdiff_train_weights* w, const stdiff_train_acts* a, const int64_t* t,
                             void* side_stream, void* stream);
long long stdiff_dit_backward_parts(const stdiff_train_weights* w);   /* floats of stdiff_grad_ranges.parts; < 0: bad w */
/* after_bucket (may be NULL) is called ON THE HOST, in order, with i = 0 (final layer), 1 .. depth (blocks depth-1 .. 0),
 * depth+1 (embedding layers) once everything that finishes that range of gradients has been enqueued on `stream`
 * (side-stream jobs joined): the data-parallel step issues the range's all-reduce from it (stDiff_train.py has no
 * counterpart; torch DDP's bucket hooks do the same). */
typedef void (*stdiff_bucket_fn)(int bucket, void* user);
int stdiff_dit_backward(const stdiff_train_weights* w, const stdiff_train_acts* a, void* side_stream,
                        stdiff_bucket_fn after_bucket, void* user, void* stream);

/* ---------------------------------------------------------------- hipGraph capture of a reverse step
 * begin/end bracket any sequence of the calls above on `stream`; launch replays it (model/sample.py:67-95 loop body
 * captured once, replayed num_step times with t read from device memory). */
int stdiff_graph_begin(void* stream);
int stdiff_graph_end(void* stream, void** graph_exec_out);
int stdiff_graph_launch(void* graph_exec, void* stream);
int stdiff_graph_destroy(void* graph_exec);

#ifdef __cplusplus
}
#endif
#endif
