// Host-side orchestration of one training pass of the denoiser (stDiff_train.py:62-77 minus loss and optimizer, which
// are kernels of their own): the forward with saved activations and the hand-written backward, each as ONE enqueue.
// ~110 launches forward, ~190 backward at depth 6; issued from C++ the host spends a few hundred microseconds on them
// instead of one interpreter round trip per launch, which is what bounds an eager (data-parallel) step at 512 rows.
//
// Weight-gradient fork: dW = dY^T X and db = colsum(dY) are leaves of the backward graph.  With a side stream they are
// issued there (own split-K scratch), one job in flight next to the dX chain, joined one job late; the forward's adaLN
// projections (functions of SiLU(c) alone) go there as one chain.  Fork / join use a small pool of HIP events that the
// library creates once per device (events hold no device memory); under stream capture they become graph edges.
// No device memory is allocated here; every buffer comes from the caller.  Not thread-safe per device.
#include <initializer_list>
#include <utility>
#include <mutex>

#include "common.h"

#define TRY(expr)                           \
    do {                                    \
        int rc__ = (expr);                  \
        if (rc__ != STDIFF_OK) return rc__; \
    } while (0)
#define TRYH(expr)                                  \
    do {                                            \
        hipError_t e__ = (expr);                    \
        if (e__ != hipSuccess) return (int)e__;     \
    } while (0)

namespace {
constexpr int MAX_DEV = 16, MAX_DEPTH = 60, N_EV = MAX_DEPTH + 4;
// The ONE piece of process-wide state of this file: the per-device pool of fork / join events (no device memory; made
// on a device's first training enqueue, kept until the process ends).  Creation is serialised by a mutex, so two host
// threads that enqueue on the same device for the first time find one pool; USING the pool is per device and not
// thread-safe, like the rest of an enqueue (include/stdiff_hip.h: one enqueue per device at a time).
hipEvent_t g_events[MAX_DEV][N_EV];
bool g_events_ready[MAX_DEV];
std::mutex g_events_mutex;

int events_for_device(hipEvent_t** out) {
    int dev = 0;
    TRYH(hipGetDevice(&dev));
    if (dev < 0 || dev >= MAX_DEV) return STDIFF_ERR_ARG;
    std::lock_guard<std::mutex> lock(g_events_mutex);
    if (!g_events_ready[dev]) {
        for (int i = 0; i < N_EV; ++i) TRYH(hipEventCreateWithFlags(&g_events[dev][i], hipEventDisableTiming));
        g_events_ready[dev] = true;
    }
    *out = g_events[dev];
    return STDIFF_OK;
}

inline const void* col(const void* base, int dtype, size_t elems) {
    return (const unsigned char*)base + elems * dtype_size(dtype);
}
inline void* col(void* base, int dtype, size_t elems) { return (unsigned char*)base + elems * dtype_size(dtype); }
inline int round_up(int a, int b) { return (a + b - 1) / b * b; }

struct Ctx {
    const stdiff_train_weights* w;
    const stdiff_train_acts* a;
    hipStream_t main, side;
    hipEvent_t* ev;              // [0] ready, [1..2] done (alternating), [3..] the forward's adaLN chain
    int done_i = 0;
    bool have_prev = false, forking = false;

    void* ws(bool on_side) const { return on_side ? a->splitk_ws_side : a->splitk_ws; }
    long long ws_bytes(bool on_side) const { return on_side ? a->splitk_side_bytes : a->splitk_bytes; }
    hipStream_t st(bool on_side) const { return on_side ? side : main; }

    // C[M,N] = epi(A[M,K] W[N,K]^T + b), everything row-major
    int lin(bool on_side, int M, int N, int K, const void* A, int lda, const stdiff_train_linear& l, int ldw, void* C,
            int ldc, int cdt, int epi = STDIFF_EPI_BIAS, const void* gate = nullptr, int ldgate = 0,
            const float* res = nullptr, int ldres = 0, void* C2 = nullptr, int ldc2 = 0,
            stdiff_split_result* defer = nullptr) const {
        stdiff_gemm_args g = {};
        g.defer = a->no_defer ? nullptr : defer;
        g.M = M; g.N = N; g.K = K; g.A = A; g.lda = lda; g.a_dtype = w->dtype; g.W = l.w; g.ldw = ldw; g.bias = l.b;
        g.C = C; g.ldc = ldc; g.c_dtype = cdt; g.epi = epi; g.gate = gate; g.ldgate = ldgate; g.res = res; g.ldres = ldres;
        g.C2 = C2; g.ldc2 = ldc2; g.workspace = ws(on_side); g.workspace_bytes = ws_bytes(on_side);
        return stdiff_gemm(&g, st(on_side));
    }
    // out[M, n_in] (+)= dY[M, k_pad] W, W = [k_valid rows (out features)][n_in] read K-major
    int dX(bool on_side, int M, const void* dY, int lddy, int k_pad, const stdiff_train_linear& l, int k_valid, int n_in,
           void* out, int ldo, int odt, int epi = STDIFF_EPI_BIAS, const void* side_in = nullptr, int ldside = 0,
           stdiff_split_result* defer = nullptr) const {
        stdiff_gemm_args g = {};
        g.defer = a->no_defer ? nullptr : defer;
        g.gate = side_in; g.ldgate = ldside;
        g.M = M; g.N = n_in; g.K = k_pad; g.A = dY; g.lda = lddy; g.a_dtype = w->dtype; g.W = l.w; g.ldw = n_in;
        g.C = out; g.ldc = ldo; g.c_dtype = odt; g.epi = epi; g.w_layout = STDIFF_K_MAJOR; g.k_valid = k_valid;
        g.workspace = ws(on_side); g.workspace_bytes = ws_bytes(on_side);
        return stdiff_gemm(&g, st(on_side));
    }
    // gw[m, n] = dY[:, :m]^T X[:, :n] ; gb[m] = colsum(dY[:, :m])      (dY [C, >= m], X [C, >= n])
    int dW_enqueue(bool on_side, const stdiff_train_linear& l, int m, int n, const void* dY, int lddy, int dy_cols,
                   const void* X, int ldx, int x_cols) const {
        const int dt = w->dtype, C = a->C;
        hipStream_t s = st(on_side);
        if (m % 8 || n % 8) {       // odd gene counts: 16-byte column pieces do not exist, go through transposed copies
            void* bufY = a->tr[on_side ? 1 : 0][0];
            void* bufX = a->tr[on_side ? 1 : 0][1];
            if (!bufY || !bufX) return STDIFF_ERR_ARG;
            const int Cp = round_up(C, 64);
            TRY(stdiff_transpose(dY, lddy, bufY, Cp, C, dy_cols, Cp, dt, s));
            TRY(stdiff_transpose(X, ldx, bufX, Cp, C, x_cols, Cp, dt, s));
            stdiff_gemm_args g = {};
            g.M = m; g.N = n; g.K = Cp; g.A = bufY; g.lda = Cp; g.a_dtype = dt; g.W = bufX; g.ldw = Cp;
            g.C = l.gw; g.ldc = n; g.c_dtype = STDIFF_F32; g.epi = STDIFF_EPI_BIAS;
            g.workspace = ws(on_side); g.workspace_bytes = ws_bytes(on_side);
            TRY(stdiff_gemm(&g, s));
        } else {
            stdiff_gemm_args g = {};
            g.M = m; g.N = n; g.K = round_up(C, dt == STDIFF_BF16 ? 64 : 16); g.k_valid = C;
            g.A = dY; g.lda = lddy; g.a_dtype = dt; g.a_layout = STDIFF_K_MAJOR;
            g.W = X; g.ldw = ldx; g.w_layout = STDIFF_K_MAJOR;
            g.C = l.gw; g.ldc = n; g.c_dtype = STDIFF_F32; g.epi = STDIFF_EPI_BIAS;
            g.workspace = ws(on_side); g.workspace_bytes = ws_bytes(on_side);
            TRY(stdiff_gemm(&g, s));
        }
        return stdiff_colsum(dY, lddy, dt, l.gb, 0, C, m, s);
    }
    // One weight-gradient job; `then` (may be empty) is more work on the same inputs that nothing on the dX chain needs.
    // Fork: the side stream starts when dY is final.  Join, one job late: from here on the caller's stream waits for the
    // PREVIOUS job -- between two jobs nothing on the caller's stream rewrites the dY / X of the job just issued
    // (dmod alternates between two buffers for exactly that reason).
    template <typename Then>
    int dW(const stdiff_train_linear& l, int m, int n, const void* dY, int lddy, int dy_cols, const void* X, int ldx,
           int x_cols, Then then) {
        if (!forking) {
            TRY(dW_enqueue(false, l, m, n, dY, lddy, dy_cols, X, ldx, x_cols));
            return then(false);
        }
        TRYH(hipEventRecord(ev[0], main));
        TRYH(hipStreamWaitEvent(side, ev[0], 0));
        TRY(dW_enqueue(true, l, m, n, dY, lddy, dy_cols, X, ldx, x_cols));
        TRY(then(true));
        hipEvent_t done = ev[1 + done_i];
        TRYH(hipEventRecord(done, side));
        if (have_prev) TRYH(hipStreamWaitEvent(main, ev[1 + (done_i ^ 1)], 0));
        have_prev = true;
        done_i ^= 1;
        return STDIFF_OK;
    }
    int dW(const stdiff_train_linear& l, int m, int n, const void* dY, int lddy, int dy_cols, const void* X, int ldx,
           int x_cols) {
        return dW(l, m, n, dY, lddy, dy_cols, X, ldx, x_cols, [](bool) { return (int)STDIFF_OK; });
    }
    // Several weight gradients in ONE launch (stdiff_dw_group) + `then`, forked like dW() but NOT joined here: the
    // grouped backward keeps two sets of dY buffers and calls join() itself where the older set is first rewritten.
    template <typename Then>
    int group(const stdiff_dw_group_args& g, Then then) {
        if (!forking) {
            TRY(stdiff_dw_group(&g, main));
            return then(false);
        }
        TRYH(hipEventRecord(ev[0], main));
        TRYH(hipStreamWaitEvent(side, ev[0], 0));
        TRY(stdiff_dw_group(&g, side));
        TRY(then(true));
        TRYH(hipEventRecord(ev[1 + done_i], side));
        have_prev = true;
        done_i ^= 1;
        return STDIFF_OK;
    }
    // gradient-norm contribution of bucket i (stdiff_train_acts.grad_ranges), on the stream that finished the bucket
    // in_group: the weight-gradient groups leave the squares of what they write in slices of grad_ranges->parts (no
    // pass over the gradients); sum_parts() adds them to *sumsq once the last group has been joined
    bool norm_in_group() const { return a->grad_ranges && a->grad_ranges->sumsq && a->grad_ranges->in_group && w->G % 8 == 0; }
    long long parts_used = 0, parts_last = 0;
    int take_parts(stdiff_dw_group_args& g) {
        if (!norm_in_group() || a->grad_ranges->in_group == 2) return STDIFF_OK;      // 2: timing probe, no norm at all
        const long long n = stdiff_dw_group_parts(&g);
        STDIFF_CHECK_ARG(n >= 0 && a->grad_ranges->parts && parts_used + n <= a->grad_ranges->parts_count);
        g.sq_parts = a->grad_ranges->parts + parts_used;
        parts_last = n;
        parts_used += n;
        return STDIFF_OK;
    }
    // *sumsq += the parts of the group launched last, on that group's stream (groups run in stream order on the side
    // stream, the embedding layers' group after the join: one fixed summation order)
    int sum_parts(bool on_side) {
        if (parts_last == 0) return STDIFF_OK;
        const long long n = parts_last;
        parts_last = 0;
        return stdiff_sum_parts(a->grad_ranges->parts + parts_used - n, n, a->grad_ranges->sumsq, 1, st(on_side));
    }
    int norm(bool on_side, int bucket) const {
        const stdiff_grad_ranges* r = a->grad_ranges;
        if (norm_in_group()) return STDIFF_OK;
        if (!r || !r->sumsq || bucket < 0 || bucket >= 64 || !r->lo[bucket] || r->count[bucket] <= 0) return STDIFF_OK;
        return stdiff_grad_sumsq(r->lo[bucket], r->count[bucket], r->sumsq, st(on_side));
    }
    static stdiff_dw_problem prob(const stdiff_train_linear& l, int m, int n, const void* dY, int lddy, const void* X,
                                  int ldx) {
        stdiff_dw_problem q = {};
        q.M = m; q.N = n; q.dY = dY; q.lddy = lddy; q.X = X; q.ldx = ldx; q.gw = l.gw; q.ldgw = n; q.gb = l.gb;
        return q;
    }
    // the caller's stream waits for every weight gradient issued so far
    int join() {
        if (forking && have_prev) TRYH(hipStreamWaitEvent(main, ev[1 + (done_i ^ 1)], 0));
        return STDIFF_OK;
    }
    // A residual-stream tensor whose producing GEMM may have left it as split-K partial products (stdiff_gemm's defer):
    // the LayerNorm that consumes it then sums the slices and applies the producer's bias / gated residual itself.
    struct Pend {
        stdiff_split_result sp = {};
        const float* bias = nullptr; const void* gate = nullptr; int ldgate = 0; const float* res = nullptr; int ldres = 0;
        void* v = nullptr; int ldv = 0;
    };
    int ln(const Pend& pd, float* x, int ldx, const void* shift, const void* scale, int ldmod, void* y, float* mean,
           float* rstd) const {
        const int dt = w->dtype, H = w->H;
        if (pd.sp.slices > 1)
            return stdiff_ln_modulate_split(&pd.sp, pd.bias, pd.gate, pd.ldgate, pd.res, pd.ldres, x, ldx, pd.v, pd.ldv,
                                            shift, scale, ldmod, dt, y, H, dt, mean, rstd, a->C, H, main);
        return stdiff_ln_modulate(x, ldx, shift, scale, ldmod, dt, y, H, dt, mean, rstd, a->C, H, main);
    }
    // LayerNorm backward (+ the gate product of the branch below when `gate` is given) of a dy that may be deferred
    int ln_bwd(const stdiff_split_result& sp, const void* dy, const float* x, const float* mean, const float* rstd,
               const void* scale, int ldmod, int accumulate, void* dshift, void* dscale, int lddmod, const void* gate,
               int ldgate, const void* branch, void* dbranch, void* dgate, int lddg) const {
        const int dt = w->dtype, H = w->H, C = a->C;
        if (sp.slices > 1)
            return stdiff_ln_modulate_bwd_split(&sp, dt, x, H, mean, rstd, scale, ldmod, a->dh, H, accumulate, dshift, dscale,
                                                lddmod, gate, ldgate, branch, H, dbranch, H, dgate, lddg, C, H, main);
        if (gate)
            return stdiff_ln_modulate_bwd_gate(dy, H, dt, x, H, mean, rstd, scale, ldmod, a->dh, H, accumulate, dshift,
                                               dscale, lddmod, gate, ldgate, branch, H, dbranch, H, dgate, lddg, C, H, main);
        return stdiff_ln_modulate_bwd(dy, H, dt, x, H, mean, rstd, scale, ldmod, a->dh, H, accumulate, dshift, dscale,
                                      lddmod, C, H, main);
    }
};

int check_common(const stdiff_train_weights* w, const stdiff_train_acts* a) {
    STDIFF_CHECK_ARG(w && a && w->blocks && a->blocks && w->freqs && a->cu_rows);
    STDIFF_CHECK_ARG(w->dtype == STDIFF_F32 || w->dtype == STDIFF_BF16);
    STDIFF_CHECK_ARG(w->depth >= 0 && w->depth <= MAX_DEPTH && a->C >= 0);
    if (w->heads <= 0 || w->H % w->heads || w->H % 64 || w->Gp % 64 || w->Gp < w->G) return STDIFF_ERR_SHAPE;
    return STDIFF_OK;
}

// The backward with ONE weight-gradient launch per DiT block (stdiff_dw_group: five products + five bias gradients,
// no K split, no reduce pass) instead of five jobs of three launches.  A block's dY operands must then stay live
// until its group has run, while the dX chain is already one block further: dbr (MLP branch), dbr (attention
// branch), dupre, dqkv and dmod exist twice and alternate with the block index; the chain waits for the group of
// block k + 1 just before block k's LayerNorm-1 backward -- the first launch that rewrites that group's buffers --
// i.e. a whole block after it was forked.  after_bucket(i) is called at those wait points.
int backward_grouped(Ctx& c, stdiff_bucket_fn after_bucket, void* user) {
    const stdiff_train_weights* w = c.w;
    const stdiff_train_acts* a = c.a;
    hipStream_t stream = c.main;
    const int dt = w->dtype, H = w->H, G = w->G, Gp = w->Gp, D = w->depth, C = a->C, d = H / w->heads;
    void* const dbrm[2] = {a->dbr, a->dbr_b[0]};
    void* const dbra[2] = {a->dbr_b[1], a->dbr_b[2]};
    void* const dupre[2] = {a->dupre, a->du};
    void* const dqkv[2] = {a->dqkv, a->dqkv_b};
    const bool odd_genes = G % 8 != 0;          // in / cond / out layer: no 16-byte column pieces -> the one-job path
    // ---- final layer (FinalLayer.forward, stDiff_model.py:135-141)
    if (odd_genes) TRY(c.dW(w->out, G, H, a->deps, Gp, Gp, a->xnf, H, H));
    // every input-gradient GEMM whose consumer is the next launch on this stream may leave its result as split-K partial
    // products (`sp`): the consumer sums them (stdiff_*_split) and the reduce launch disappears from the chain
    stdiff_split_result sp = {};
    TRY(c.dX(false, C, a->deps, Gp, Gp, w->out, G, H, a->dxn, H, dt, STDIFF_EPI_BIAS, nullptr, 0, &sp));
    if (D > 0) {
        const stdiff_train_block_acts& sl = a->blocks[D - 1];
        const int ql = (D - 1) & 1;
        TRY(c.ln_bwd(sp, a->dxn, a->h_last, a->meanf, a->rstdf, col(a->fmod, dt, H), 2 * H, 0, col(a->dfmod, dt, 0),
                     col(a->dfmod, dt, H), 2 * H, col(sl.mod, dt, 5 * H), 6 * H, sl.pm, dbrm[ql],
                     col(a->dmod[ql], dt, 5 * H), 6 * H));
    } else {
        TRY(c.ln_bwd(sp, a->dxn, a->h_last, a->meanf, a->rstdf, col(a->fmod, dt, H), 2 * H, 0, col(a->dfmod, dt, 0),
                     col(a->dfmod, dt, H), 2 * H, nullptr, 0, nullptr, nullptr, nullptr, 0));
    }
    {
        stdiff_dw_group_args g = {};
        g.C = C; g.dtype = dt;
        g.p[g.n++] = Ctx::prob(w->fada, 2 * H, H, a->dfmod, 2 * H, a->s, H);
        if (!odd_genes) g.p[g.n++] = Ctx::prob(w->out, G, H, a->deps, Gp, a->xnf, H);
        // d loss / d s accumulates over all adaLN projections on the stream that runs the groups, in issue order
        TRY(c.take_parts(g));
        TRY(c.group(g, [&](bool sd) {
            TRY(c.norm(sd, 0));
            TRY(c.sum_parts(sd));
            return c.dX(sd, C, a->dfmod, 2 * H, 2 * H, w->fada, 2 * H, H, a->ds32, H, STDIFF_F32);
        }));
    }
    for (int k = D - 1; k >= 0; --k) {
        const stdiff_train_block& b = w->blocks[k];
        const stdiff_train_block_acts& s = a->blocks[k];
        const int q = k & 1;
        void* dmod = a->dmod[q];
        // MLP branch (DiTblock.forward, stDiff_model.py:116); GELU backward in the fc2 dX epilogue
        TRY(c.dX(false, C, dbrm[q], H, H, b.fc2, H, 4 * H, dupre[q], 4 * H, dt, STDIFF_EPI_GELU_BWD, s.upre, 4 * H));
        TRY(c.dX(false, C, dupre[q], 4 * H, 4 * H, b.fc1, 4 * H, H, a->dxn, H, dt, STDIFF_EPI_BIAS, nullptr, 0, &sp));
        // LayerNorm 2 backward + the attention branch's gate (:114)
        TRY(c.ln_bwd(sp, a->dxn, s.hmid, s.mean2, s.rstd2, col(s.mod, dt, 4 * H), 6 * H, 1, col(dmod, dt, 3 * H),
                     col(dmod, dt, 4 * H), 6 * H, col(s.mod, dt, 2 * H), 6 * H, s.pa, dbra[q], col(dmod, dt, 2 * H), 6 * H));
        TRY(c.dX(false, C, dbra[q], H, H, b.proj, H, H, a->do_, H, dt, STDIFF_EPI_BIAS, nullptr, 0, &sp));
        if (sp.slices > 1)
            TRY(stdiff_attention_bwd_split(s.qkv, 3 * H, s.o, H, &sp, a->do_, H, s.lse, a->delta, dqkv[q], 3 * H, a->cu_rows,
                                           1, C, C, w->heads, d, dt, stream));
        else
            TRY(stdiff_attention_bwd(s.qkv, 3 * H, s.o, H, a->do_, H, s.lse, a->delta, dqkv[q], 3 * H, a->cu_rows, 1, C, C,
                                     w->heads, d, dt, stream));
        TRY(c.dX(false, C, dqkv[q], 3 * H, 3 * H, b.qkv, 3 * H, H, a->dxn, H, dt, STDIFF_EPI_BIAS, nullptr, 0, &sp));
        // the group forked a block ago (block k + 1, or the final layer's) has read the buffers of the other parity
        TRY(c.join());
        if (after_bucket) after_bucket(D - 1 - k, user);
        if (k > 0) {        // LayerNorm 1 backward + the MLP gate of the block below (the other parity's buffers)
            const stdiff_train_block_acts& sb = a->blocks[k - 1];
            TRY(c.ln_bwd(sp, a->dxn, s.h, s.mean1, s.rstd1, col(s.mod, dt, H), 6 * H, 1, col(dmod, dt, 0), col(dmod, dt, H),
                         6 * H, col(sb.mod, dt, 5 * H), 6 * H, sb.pm, dbrm[q ^ 1], col(a->dmod[q ^ 1], dt, 5 * H), 6 * H));
        } else {
            TRY(c.ln_bwd(sp, a->dxn, s.h, s.mean1, s.rstd1, col(s.mod, dt, H), 6 * H, 1, col(dmod, dt, 0), col(dmod, dt, H),
                         6 * H, nullptr, 0, nullptr, nullptr, nullptr, 0));
        }
        stdiff_dw_group_args g = {};
        g.C = C; g.dtype = dt;
        g.p[g.n++] = Ctx::prob(b.ada, 6 * H, H, dmod, 6 * H, a->s, H);
        g.p[g.n++] = Ctx::prob(b.fc1, 4 * H, H, dupre[q], 4 * H, s.xn2, H);
        g.p[g.n++] = Ctx::prob(b.fc2, H, 4 * H, dbrm[q], H, s.u, 4 * H);
        g.p[g.n++] = Ctx::prob(b.qkv, 3 * H, H, dqkv[q], 3 * H, s.xn1, H);
        g.p[g.n++] = Ctx::prob(b.proj, H, H, dbra[q], H, s.o, H);
        TRY(c.take_parts(g));
        TRY(c.group(g, [&](bool sd) {
            TRY(c.norm(sd, D - k));
            TRY(c.sum_parts(sd));
            TRY(c.dX(sd, C, dmod, 6 * H, 6 * H, b.ada, 6 * H, H, a->ds32, H, STDIFF_F32, STDIFF_EPI_ACCUM));
            // block 0: d loss / d h0 is final since the LayerNorm-1 backward above; its operand copy (in_layer's weight
            // gradient reads it) is made here, beside the chain
            if (k == 0) TRY(stdiff_pack_rows(a->dh, H, a->dhT, H, dt, C, H, H, c.st(sd)));
            return (int)STDIFF_OK;
        }));
    }
    TRY(c.join());                                 // the last group: ds32 is complete
    if (after_bucket) after_bucket(D, user);
    // ---- embeddings: h0 = in_layer(x); c = time_emb(t) + cond_layer(y); s = silu(c)   (:231-239)
    if (D == 0) TRY(stdiff_pack_rows(a->dh, H, a->dhT, H, dt, C, H, H, stream));
    // d loss / d c straight into the operand format (the fp32 copy a->dc had no other reader: one launch less)
    TRY(stdiff_silu_bwd(a->ds32, H, STDIFF_F32, a->c, H, STDIFF_F32, a->dcT, H, dt, 0, C, H, stream));
    TRY(c.dX(false, C, a->dcT, H, H, w->t2, H, H, a->de1, H, dt));
    TRY(stdiff_silu_bwd(a->de1, H, dt, a->a1, H, dt, a->da1, H, dt, 0, C, H, stream));
    {
        stdiff_dw_group_args g = {};
        g.C = C; g.dtype = dt;
        g.p[g.n++] = Ctx::prob(w->t2, H, H, a->dcT, H, a->e1, H);
        g.p[g.n++] = Ctx::prob(w->t0, H, 256, a->da1, H, a->e0, 256);
        if (!odd_genes) {
            g.p[g.n++] = Ctx::prob(w->in, H, G, a->dhT, H, a->xp, Gp);
            g.p[g.n++] = Ctx::prob(w->cond, H, G, a->dcT, H, a->yp, Gp);
        }
        TRY(c.take_parts(g));
        TRY(stdiff_dw_group(&g, stream));
        if (odd_genes) {
            TRY(c.dW_enqueue(false, w->in, H, G, a->dhT, H, H, a->xp, Gp, Gp));
            TRY(c.dW_enqueue(false, w->cond, H, G, a->dcT, H, H, a->yp, Gp, Gp));
        }
        TRY(c.norm(false, D + 1));
        TRY(c.sum_parts(false));
    }
    if (after_bucket) after_bucket(D + 1, user);
    return STDIFF_OK;
}
}  // namespace

// DiT_stDiff.forward (stDiff_model.py:231-243) on one batch of a->C cells (one attention chunk), keeping what the
// backward needs.  a->xp / a->yp hold the packed inputs; the result is a->eps fp32 [C, G].
extern "C" int stdiff_dit_train_forward(const stdiff_train_weights* w, const stdiff_train_acts* a, const int64_t* t,
                                        void* side_stream, void* stream) {
    TRY(check_common(w, a));
    STDIFF_CHECK_ARG(t && a->xp && a->yp && a->e0 && a->a1 && a->e1 && a->temb && a->c && a->s && a->h_last && a->fmod &&
                     a->xnf && a->meanf && a->rstdf && a->eps);
    const int dt = w->dtype, H = w->H, G = w->G, Gp = w->Gp, D = w->depth, C = a->C, d = H / w->heads;
    if (C == 0) return STDIFF_OK;
    Ctx c{w, a, (hipStream_t)stream, (hipStream_t)side_stream, nullptr};
    const bool fork = side_stream != nullptr && side_stream != stream;
    if (fork) {
        STDIFF_CHECK_ARG(a->splitk_ws_side || !a->splitk_ws);
        TRY(events_for_device(&c.ev));
    }
    // parameters updated range by range on another stream (stdiff_adamw_range): wait where a range is first used
    auto ready = [&](hipStream_t st, int range) -> int {
        if (a->param_ready && a->param_ready[range]) TRYH(hipStreamWaitEvent(st, (hipEvent_t)a->param_ready[range], 0));
        return STDIFF_OK;
    };
    // t -> sinusoidal embedding -> time MLP (stDiff_model.py:52-80), c = time_emb(t) + cond_layer(y) (:237), s = silu(c)
    TRY(stdiff_timestep_embedding(t, w->freqs, a->e0, 256, dt, C, stream));
    TRY(ready(c.main, 0));
    TRY(c.lin(false, C, H, 256, a->e0, 256, w->t0, 256, a->e1, H, dt, STDIFF_EPI_SILU, nullptr, 0, nullptr, 0, a->a1, H));
    TRY(c.lin(false, C, H, H, a->e1, H, w->t2, H, a->temb, H, STDIFF_F32));
    TRY(c.lin(false, C, H, Gp, a->yp, Gp, w->cond, Gp, a->c, H, STDIFF_F32, STDIFF_EPI_ADD, nullptr, 0, a->temb, H));
    TRY(stdiff_silu_cond(nullptr, nullptr, a->c, H, a->s, H, dt, C, H, stream));
    // every adaLN projection depends on s only: with a side stream they run there as one chain, and the block chain
    // waits for mod[k] where it first needs it
    if (fork) {
        TRYH(hipEventRecord(c.ev[0], c.main));
        TRYH(hipStreamWaitEvent(c.side, c.ev[0], 0));
        for (int k = 0; k <= D; ++k) {
            TRY(ready(c.side, 1 + 4 * k));             // k == D: the final layer's slot
            if (k < D) TRY(c.lin(true, C, 6 * H, H, a->s, H, w->blocks[k].ada, H, a->blocks[k].mod, 6 * H, dt));
            else TRY(c.lin(true, C, 2 * H, H, a->s, H, w->fada, H, a->fmod, 2 * H, dt));
            TRYH(hipEventRecord(c.ev[3 + k], c.side));
        }
    }
    // `pend`: the block input may still lie as split-K partial products of the GEMM that produces it (in_layer, then
    // each block's fc2): the LayerNorm that reads it first sums them (one launch less on the chain per deferral)
    // (only with the side stream: without it the block's adaLN GEMM sits between producer and consumer on this stream
    // and may use the same split-K workspace)
    Ctx::Pend pend;
    pend.bias = w->in.b;
    TRY(c.lin(false, C, H, Gp, a->xp, Gp, w->in, Gp, D ? a->blocks[0].h : a->h_last, H, STDIFF_F32, STDIFF_EPI_BIAS, nullptr,
              0, nullptr, 0, nullptr, 0, fork ? &pend.sp : nullptr));
    for (int k = 0; k < D; ++k) {
        const stdiff_train_block& b = w->blocks[k];
        const stdiff_train_block_acts& s = a->blocks[k];
        float* h_next = k + 1 < D ? a->blocks[k + 1].h : a->h_last;
        STDIFF_CHECK_ARG(s.h && s.hmid && s.mod && s.xn1 && s.xn2 && s.qkv && s.o && s.lse && s.pa && s.pm && s.upre &&
                         s.u && s.mean1 && s.rstd1 && s.mean2 && s.rstd2);
        // with the side stream, mod[k]'s event already orders this stream behind param_ready[1 + k] (the side stream waited
        // for it before the adaLN GEMM): a second cross-stream edge per block would only cost its latency
        if (fork) {
            TRYH(hipStreamWaitEvent(c.main, c.ev[3 + k], 0));
        } else {
            TRY(ready(c.main, 1 + 4 * k));
            TRY(c.lin(false, C, 6 * H, H, a->s, H, b.ada, H, s.mod, 6 * H, dt));
        }
        TRY(c.ln(pend, s.h, H, col(s.mod, dt, 0), col(s.mod, dt, H), 6 * H, s.xn1, s.mean1, s.rstd1));
        TRY(ready(c.main, 1 + 4 * k + 1));              // (NULL unless the block is updated in pieces)
        TRY(c.lin(false, C, 3 * H, H, s.xn1, H, b.qkv, H, s.qkv, 3 * H, dt));
        TRY(stdiff_attention_fwd(s.qkv, 3 * H, s.o, H, s.lse, a->cu_rows, 1, C, w->heads, d, dt, stream));
        Ctx::Pend mid;
        mid.bias = b.proj.b; mid.gate = col(s.mod, dt, 2 * H); mid.ldgate = 6 * H; mid.res = s.h; mid.ldres = H;
        mid.v = s.pa; mid.ldv = H;
        TRY(c.lin(false, C, H, H, s.o, H, b.proj, H, s.hmid, H, STDIFF_F32, STDIFF_EPI_GATE_RES, col(s.mod, dt, 2 * H),
                  6 * H, s.h, H, s.pa, H, &mid.sp));
        TRY(c.ln(mid, s.hmid, H, col(s.mod, dt, 3 * H), col(s.mod, dt, 4 * H), 6 * H, s.xn2, s.mean2, s.rstd2));
        TRY(ready(c.main, 1 + 4 * k + 2));
        TRY(c.lin(false, C, 4 * H, H, s.xn2, H, b.fc1, H, s.u, 4 * H, dt, STDIFF_EPI_GELU_SAVE, nullptr, 0, nullptr, 0,
                  s.upre, 4 * H));
        TRY(ready(c.main, 1 + 4 * k + 3));
        pend = Ctx::Pend();
        pend.bias = b.fc2.b; pend.gate = col(s.mod, dt, 5 * H); pend.ldgate = 6 * H; pend.res = s.hmid; pend.ldres = H;
        pend.v = s.pm; pend.ldv = H;
        TRY(c.lin(false, C, H, 4 * H, s.u, 4 * H, b.fc2, 4 * H, h_next, H, STDIFF_F32, STDIFF_EPI_GATE_RES,
                  col(s.mod, dt, 5 * H), 6 * H, s.hmid, H, s.pm, H, fork ? &pend.sp : nullptr));
    }
    if (fork) {
        TRYH(hipStreamWaitEvent(c.main, c.ev[3 + D], 0));
    } else {
        TRY(ready(c.main, 1 + 4 * D));
        TRY(c.lin(false, C, 2 * H, H, a->s, H, w->fada, H, a->fmod, 2 * H, dt));
    }
    TRY(c.ln(pend, a->h_last, H, col(a->fmod, dt, 0), col(a->fmod, dt, H), 2 * H, a->xnf, a->meanf, a->rstdf));
    return c.lin(false, C, G, H, a->xnf, H, w->out, H, a->eps, G, STDIFF_F32);
}

// Backward of the above: a->deps holds d loss / d eps as the packed operand [C, Gp]; every gw / gb of `w` is
// overwritten.  after_bucket(i, user), when given, is called on the host when the launches that finish the i-th
// gradient range (final layer; block depth-1 .. 0; embedding layers) have been enqueued on `stream` -- the
// data-parallel step issues that range's all-reduce from it.
// floats of stdiff_grad_ranges.parts that backward_grouped() hands to its stdiff_dw_group launches (same groups, same order)
extern "C" long long stdiff_dit_backward_parts(const stdiff_train_weights* w) {
    if (!w || w->H <= 0 || w->G <= 0 || w->depth < 0 || w->depth > MAX_DEPTH) return -1;
    if (w->G % 8 != 0) return 0;
    const int H = w->H, G = w->G;
    auto parts = [](std::initializer_list<std::pair<int, int>> mn) {
        stdiff_dw_group_args g = {};
        for (auto& q : mn) { g.p[g.n].M = q.first; g.p[g.n].N = q.second; ++g.n; }
        return stdiff_dw_group_parts(&g);
    };
    return parts({{2 * H, H}, {G, H}}) + w->depth * parts({{6 * H, H}, {4 * H, H}, {H, 4 * H}, {3 * H, H}, {H, H}}) +
           parts({{H, H}, {H, 256}, {H, G}, {H, G}});
}

extern "C" int stdiff_dit_backward(const stdiff_train_weights* w, const stdiff_train_acts* a, void* side_stream,
                                   stdiff_bucket_fn after_bucket, void* user, void* stream) {
    TRY(check_common(w, a));
    STDIFF_CHECK_ARG(a->deps && a->dh && a->dxn && a->dmod[0] && a->dmod[1] && a->dfmod && a->dbr && a->du && a->dupre &&
                     a->do_ && a->dqkv && a->delta && a->ds32 && a->dc && a->dcT && a->de1 && a->da1 && a->dhT);
    const int dt = w->dtype, H = w->H, G = w->G, Gp = w->Gp, D = w->depth, C = a->C, d = H / w->heads;
    if (C == 0) return STDIFF_OK;
    Ctx c{w, a, (hipStream_t)stream, (hipStream_t)side_stream, nullptr};
    c.forking = side_stream != nullptr && side_stream != stream;
    if (c.forking) {
        STDIFF_CHECK_ARG(a->splitk_ws_side || !a->splitk_ws);
        TRY(events_for_device(&c.ev));
    }
    if (a->dbr_b[0] && a->dbr_b[1] && a->dbr_b[2] && a->dqkv_b) return backward_grouped(c, after_bucket, user);
    // ---- one job per linear layer (callers that did not provide the second set of dY buffers)
    // final layer (FinalLayer.forward, stDiff_model.py:135-141)
    TRY(c.dW(w->out, G, H, a->deps, Gp, Gp, a->xnf, H, H));
    TRY(c.dX(false, C, a->deps, Gp, Gp, w->out, G, H, a->dxn, H, dt));
    // every LayerNorm backward of the chain is followed by the gated-residual backward of the branch below it: one launch
    // (stdiff_ln_modulate_bwd_gate).  Here: the MLP branch of the last block, h_D = hmid + gate_mlp * pm (:116)
    if (D > 0) {
        const stdiff_train_block_acts& sl = a->blocks[D - 1];
        void* dml = a->dmod[(D - 1) & 1];
        TRY(stdiff_ln_modulate_bwd_gate(a->dxn, H, dt, a->h_last, H, a->meanf, a->rstdf, col(a->fmod, dt, H), 2 * H, a->dh,
                                        H, 0, col(a->dfmod, dt, 0), col(a->dfmod, dt, H), 2 * H, col(sl.mod, dt, 5 * H),
                                        6 * H, sl.pm, H, a->dbr, H, col(dml, dt, 5 * H), 6 * H, C, H, stream));
    } else {
        TRY(stdiff_ln_modulate_bwd(a->dxn, H, dt, a->h_last, H, a->meanf, a->rstdf, col(a->fmod, dt, H), 2 * H, a->dh, H,
                                   0, col(a->dfmod, dt, 0), col(a->dfmod, dt, H), 2 * H, C, H, stream));
    }
    // d loss / d s accumulates over all adaLN projections and is needed only after the last block: its GEMMs ride with
    // the adaLN weight-gradient jobs (same dY), in issue order on whichever stream runs them
    TRY(c.dW(w->fada, 2 * H, H, a->dfmod, 2 * H, 2 * H, a->s, H, H, [&](bool sd) {
        return c.dX(sd, C, a->dfmod, 2 * H, 2 * H, w->fada, 2 * H, H, a->ds32, H, STDIFF_F32);
    }));
    if (after_bucket) {
        TRY(c.join());
        after_bucket(0, user);
    }
    for (int k = D - 1; k >= 0; --k) {
        const stdiff_train_block& b = w->blocks[k];
        const stdiff_train_block_acts& s = a->blocks[k];
        void* dmod = a->dmod[k & 1];
        // MLP branch: h_{k+1} = hmid + gate_mlp * pm   (DiTblock.forward, stDiff_model.py:116): dbr = dh * gate_mlp and
        // d gate_mlp were written by the LayerNorm backward above this block (fused launch)
        TRY(c.dW(b.fc2, H, 4 * H, a->dbr, H, H, s.u, 4 * H, 4 * H));
        // d(fc2 input) with the GELU backward in the GEMM's epilogue: dupre = (dbr W_fc2) * gelu'(upre)
        TRY(c.dX(false, C, a->dbr, H, H, b.fc2, H, 4 * H, a->dupre, 4 * H, dt, STDIFF_EPI_GELU_BWD, s.upre, 4 * H));
        TRY(c.dW(b.fc1, 4 * H, H, a->dupre, 4 * H, 4 * H, s.xn2, H, H));
        TRY(c.dX(false, C, a->dupre, 4 * H, 4 * H, b.fc1, 4 * H, H, a->dxn, H, dt));
        // LayerNorm 2 backward + the attention branch's gate: hmid = h_k + gate_msa * pa   (:114)
        TRY(stdiff_ln_modulate_bwd_gate(a->dxn, H, dt, s.hmid, H, s.mean2, s.rstd2, col(s.mod, dt, 4 * H), 6 * H, a->dh, H, 1,
                                        col(dmod, dt, 3 * H), col(dmod, dt, 4 * H), 6 * H, col(s.mod, dt, 2 * H), 6 * H, s.pa,
                                        H, a->dbr, H, col(dmod, dt, 2 * H), 6 * H, C, H, stream));
        TRY(c.dW(b.proj, H, H, a->dbr, H, H, s.o, H, H));
        TRY(c.dX(false, C, a->dbr, H, H, b.proj, H, H, a->do_, H, dt));
        TRY(stdiff_attention_bwd(s.qkv, 3 * H, s.o, H, a->do_, H, s.lse, a->delta, a->dqkv, 3 * H, a->cu_rows, 1, C, C,
                                 w->heads, d, dt, stream));
        TRY(c.dW(b.qkv, 3 * H, H, a->dqkv, 3 * H, 3 * H, s.xn1, H, H));
        TRY(c.dX(false, C, a->dqkv, 3 * H, 3 * H, b.qkv, 3 * H, H, a->dxn, H, dt));
        if (k > 0) {        // LayerNorm 1 backward + the MLP gate of the block below (its dmod buffer is the other one)
            const stdiff_train_block_acts& sb = a->blocks[k - 1];
            void* dmb = a->dmod[(k - 1) & 1];
            TRY(stdiff_ln_modulate_bwd_gate(a->dxn, H, dt, s.h, H, s.mean1, s.rstd1, col(s.mod, dt, H), 6 * H, a->dh, H, 1,
                                            col(dmod, dt, 0), col(dmod, dt, H), 6 * H, col(sb.mod, dt, 5 * H), 6 * H, sb.pm,
                                            H, a->dbr, H, col(dmb, dt, 5 * H), 6 * H, C, H, stream));
        } else {
            TRY(stdiff_ln_modulate_bwd(a->dxn, H, dt, s.h, H, s.mean1, s.rstd1, col(s.mod, dt, H), 6 * H, a->dh, H, 1,
                                       col(dmod, dt, 0), col(dmod, dt, H), 6 * H, C, H, stream));
        }
        TRY(c.dW(b.ada, 6 * H, H, dmod, 6 * H, 6 * H, a->s, H, H, [&](bool sd) {
            return c.dX(sd, C, dmod, 6 * H, 6 * H, b.ada, 6 * H, H, a->ds32, H, STDIFF_F32, STDIFF_EPI_ACCUM);
        }));
        if (after_bucket) {
            TRY(c.join());
            after_bucket(D - k, user);
        }
    }
    // ---- embeddings: h0 = in_layer(x); c = time_emb(t) + cond_layer(y); s = silu(c)   (:231-239)
    TRY(stdiff_pack_rows(a->dh, H, a->dhT, H, dt, C, H, H, stream));
    TRY(c.dW(w->in, H, G, a->dhT, H, H, a->xp, Gp, Gp));
    TRY(c.join());                                 // ds32 is complete
    TRY(stdiff_silu_bwd(a->ds32, H, STDIFF_F32, a->c, H, STDIFF_F32, a->dc, H, STDIFF_F32, 0, C, H, stream));
    TRY(stdiff_pack_rows(a->dc, H, a->dcT, H, dt, C, H, H, stream));
    TRY(c.dW(w->cond, H, G, a->dcT, H, H, a->yp, Gp, Gp));
    TRY(c.dW(w->t2, H, H, a->dcT, H, H, a->e1, H, H));
    TRY(c.dX(false, C, a->dcT, H, H, w->t2, H, H, a->de1, H, dt));
    TRY(stdiff_silu_bwd(a->de1, H, dt, a->a1, H, dt, a->da1, H, dt, 0, C, H, stream));
    TRY(c.dW(w->t0, H, 256, a->da1, H, H, a->e0, 256, 256));
    TRY(c.join());                                 // every gradient is final on the caller's stream from here on
    if (after_bucket) after_bucket(D + 1, user);
    return STDIFF_OK;
}
