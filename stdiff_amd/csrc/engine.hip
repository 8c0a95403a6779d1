// Host-side orchestration: the whole-denoiser forward as one enqueue, and hipGraph capture/replay of a reverse step.
// No device memory is allocated here; every buffer comes from the caller.
#include "common.h"

extern "C" int stdiff_abi_version(void) { return STDIFF_ABI_VERSION; }

extern "C" int stdiff_h16_format(void) {
#ifdef STDIFF_H16_FP16
    return 1;
#else
    return 0;
#endif
}

extern "C" const char* stdiff_error_string(int s) {
    switch (s) {
        case STDIFF_OK: return "ok";
        case STDIFF_ERR_ARG: return "bad argument (null pointer, negative size or unknown enum)";
        case STDIFF_ERR_SHAPE: return "shape not supported by the gfx950 kernels";
        case STDIFF_ERR_ALIGN: return "pointer or leading dimension not 16-byte aligned";
        case STDIFF_ERR_STATE: return "handle in the wrong state";
        case STDIFF_ERR_SCRATCH: return "split-K problem exceeds the library's static scratch buffer (3 * M * round_up(N, 4) * 4 bytes > STDIFF_STATIC_SCRATCH_BYTES)";
        default: return s > 0 ? hipGetErrorString((hipError_t)s) : "unknown stdiff status";
    }
}

#define TRY(expr)                 \
    do {                          \
        int rc__ = (expr);        \
        if (rc__ != STDIFF_OK) return rc__; \
    } while (0)

namespace {
inline const void* col(const void* base, int dtype, size_t elems) {
    return (const unsigned char*)base + elems * dtype_size(dtype);
}
int linear(int M, int N, int K, const void* A, int lda, int dt, const void* W, int ldw, const float* b, void* C,
           int ldc, int cdt, int epi, const void* gate, int ldgate, const float* res, int ldres, void* st,
           const float* ln_parts = nullptr, int ln_nparts = 0, float* stats_out = nullptr) {
    stdiff_gemm_args g = {};
    g.M = M; g.N = N; g.K = K; g.A = A; g.lda = lda; g.a_dtype = dt; g.W = W; g.ldw = ldw; g.bias = b;
    g.C = C; g.ldc = ldc; g.c_dtype = cdt; g.epi = epi; g.gate = gate; g.ldgate = ldgate; g.res = res; g.ldres = ldres;
    g.ln_parts = ln_parts; g.ln_nparts = ln_nparts; g.stats_out = stats_out;
    return stdiff_gemm(&g, st);
}

// xn = modulate(LayerNorm(h); shift, scale) with (shift, scale) = Linear(s) computed by THIS GEMM and consumed in its
// epilogue (w: [2H, H] in the [128 shift | 128 scale] tile order).  The statistics of h are `nparts` partial moments per
// row in ws->ln_parts: written by the epilogue that produced h (nparts = H/256), or by the statistics pass here (0 -> 1).
int ln_linear(const stdiff_dit_workspace* ws, int R, int H, int dt, const void* s, const void* w, const float* b,
              int nparts, void* st) {
    if (nparts == 0) {
        TRY(stdiff_row_stats(ws->h, H, ws->ln_parts, R, H, st));
        nparts = 1;
    }
    return linear(R, 2 * H, H, s, H, dt, w, H, b, ws->xn, H, dt, STDIFF_EPI_LN_MOD, nullptr, 0, ws->h, H, st,
                  ws->ln_parts, nparts);
}
}  // namespace

// DiT_stDiff.forward body (stDiff_model.py:240-243; DiTblock.forward :110-117; FinalLayer.forward :135-141)
extern "C" int stdiff_dit_forward(const stdiff_dit_weights* w, const stdiff_dit_workspace* ws, const void* xp,
                                  const void* s, float* eps, int ldeps, void* stream) {
    STDIFF_CHECK_ARG(w && ws && xp && s && eps && w->blocks);
    STDIFF_CHECK_ARG(w->dtype == STDIFF_F32 || w->dtype == STDIFF_BF16);
    STDIFF_CHECK_ARG(ws->h && ws->xn && ws->qkv && ws->o && ws->u && ws->cu_rows);
    const int dt = w->dtype, H = w->H, R = ws->R, d = H / w->heads;
    if (w->heads <= 0 || H % w->heads || H % 64 || w->Gp % 64 || w->Gp < w->G) return STDIFF_ERR_SHAPE;
    if (R == 0) return STDIFF_OK;
    bool fused = dt == STDIFF_BF16 && H % 128 == 0 && ws->ln_parts && ws->gates && w->lnf_w && w->lnf_b;
    for (int k = 0; fused && k < w->depth; ++k)
        fused = w->blocks[k].ln1_w && w->blocks[k].ln1_b && w->blocks[k].ln2_w && w->blocks[k].ln2_b &&
                w->blocks[k].gate_w && w->blocks[k].gate_b;
    if (!fused) STDIFF_CHECK_ARG(ws->mod && ws->fmod);
    if (fused) {
        // ---- fused-LN form (include/stdiff_hip.h, STDIFF_EPI_LN_MOD).  With H a multiple of 256 the gated-residual
        // epilogues of proj / fc2 also hand the next LayerNorm its statistics (P <= 8 parts per row); otherwise, and for
        // the first LayerNorm (after in_layer), a statistics pass reads h once.
        const int P = (H % 256 == 0 && H / 256 <= 8) ? H / 256 : 0;      // the LN_MOD epilogue combines up to 8 parts
        float* so = P ? ws->ln_parts : nullptr;
        TRY(linear(R, H, w->Gp, xp, w->Gp, dt, w->in_w, w->Gp, w->in_b, ws->h, H, STDIFF_F32, STDIFF_EPI_BIAS, nullptr, 0,
                   nullptr, 0, stream));
        for (int k = 0; k < w->depth; ++k) {
            const stdiff_block_weights& b = w->blocks[k];
            // (gate_msa | gate_mlp) = the two gate slices of Linear(SiLU(c))
            TRY(linear(R, 2 * H, H, s, H, dt, b.gate_w, H, b.gate_b, ws->gates, 2 * H, dt, STDIFF_EPI_BIAS, nullptr, 0,
                       nullptr, 0, stream));
            TRY(ln_linear(ws, R, H, dt, s, b.ln1_w, b.ln1_b, k == 0 ? 0 : P, stream));
            TRY(linear(R, 3 * H, H, ws->xn, H, dt, b.qkv_w, H, b.qkv_b, ws->qkv, 3 * H, dt, STDIFF_EPI_BIAS, nullptr, 0,
                       nullptr, 0, stream));
            TRY(stdiff_attention_fwd(ws->qkv, 3 * H, ws->o, H, nullptr, ws->cu_rows, ws->n_chunks, ws->max_chunk_rows,
                                     w->heads, d, dt, stream));
            TRY(linear(R, H, H, ws->o, H, dt, b.proj_w, H, b.proj_b, ws->h, H, STDIFF_F32, STDIFF_EPI_GATE_RES,
                       col(ws->gates, dt, 0), 2 * H, ws->h, H, stream, nullptr, 0, so));
            TRY(ln_linear(ws, R, H, dt, s, b.ln2_w, b.ln2_b, P, stream));
            TRY(linear(R, 4 * H, H, ws->xn, H, dt, b.fc1_w, H, b.fc1_b, ws->u, 4 * H, dt, STDIFF_EPI_GELU, nullptr, 0,
                       nullptr, 0, stream));
            TRY(linear(R, H, 4 * H, ws->u, 4 * H, dt, b.fc2_w, 4 * H, b.fc2_b, ws->h, H, STDIFF_F32, STDIFF_EPI_GATE_RES,
                       col(ws->gates, dt, H), 2 * H, ws->h, H, stream, nullptr, 0, so));
        }
        TRY(ln_linear(ws, R, H, dt, s, w->lnf_w, w->lnf_b, w->depth > 0 ? P : 0, stream));
        TRY(linear(R, w->G, H, ws->xn, H, dt, w->out_w, H, w->out_b, eps, ldeps, STDIFF_F32, STDIFF_EPI_BIAS, nullptr, 0,
                   nullptr, 0, stream));
        return STDIFF_OK;
    }
    // h = in_layer(x)
    TRY(linear(R, H, w->Gp, xp, w->Gp, dt, w->in_w, w->Gp, w->in_b, ws->h, H, STDIFF_F32, STDIFF_EPI_BIAS, nullptr, 0,
               nullptr, 0, stream));
    for (int k = 0; k < w->depth; ++k) {
        const stdiff_block_weights& b = w->blocks[k];
        // (shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp) = Linear(SiLU(c)).chunk(6)
        TRY(linear(R, 6 * H, H, s, H, dt, b.ada_w, H, b.ada_b, ws->mod, 6 * H, dt, STDIFF_EPI_BIAS, nullptr, 0, nullptr,
                   0, stream));
        TRY(stdiff_ln_modulate(ws->h, H, col(ws->mod, dt, 0), col(ws->mod, dt, H), 6 * H, dt, ws->xn, H, dt, nullptr,
                               nullptr, R, H, stream));
        TRY(linear(R, 3 * H, H, ws->xn, H, dt, b.qkv_w, H, b.qkv_b, ws->qkv, 3 * H, dt, STDIFF_EPI_BIAS, nullptr, 0,
                   nullptr, 0, stream));
        TRY(stdiff_attention_fwd(ws->qkv, 3 * H, ws->o, H, nullptr, ws->cu_rows, ws->n_chunks, ws->max_chunk_rows,
                                 w->heads, d, dt, stream));
        // x = x + gate_msa * proj(attn)
        TRY(linear(R, H, H, ws->o, H, dt, b.proj_w, H, b.proj_b, ws->h, H, STDIFF_F32, STDIFF_EPI_GATE_RES,
                   col(ws->mod, dt, 2 * H), 6 * H, ws->h, H, stream));
        TRY(stdiff_ln_modulate(ws->h, H, col(ws->mod, dt, 3 * H), col(ws->mod, dt, 4 * H), 6 * H, dt, ws->xn, H, dt,
                               nullptr, nullptr, R, H, stream));
        TRY(linear(R, 4 * H, H, ws->xn, H, dt, b.fc1_w, H, b.fc1_b, ws->u, 4 * H, dt, STDIFF_EPI_GELU, nullptr, 0,
                   nullptr, 0, stream));
        // x = x + gate_mlp * fc2(gelu(fc1(.)))
        TRY(linear(R, H, 4 * H, ws->u, 4 * H, dt, b.fc2_w, 4 * H, b.fc2_b, ws->h, H, STDIFF_F32, STDIFF_EPI_GATE_RES,
                   col(ws->mod, dt, 5 * H), 6 * H, ws->h, H, stream));
    }
    // final layer: (shift, scale) = Linear(SiLU(c)).chunk(2); eps = Linear(modulate(LN(x)))
    TRY(linear(R, 2 * H, H, s, H, dt, w->fada_w, H, w->fada_b, ws->fmod, 2 * H, dt, STDIFF_EPI_BIAS, nullptr, 0, nullptr,
               0, stream));
    TRY(stdiff_ln_modulate(ws->h, H, col(ws->fmod, dt, 0), col(ws->fmod, dt, H), 2 * H, dt, ws->xn, H, dt, nullptr,
                           nullptr, R, H, stream));
    TRY(linear(R, w->G, H, ws->xn, H, dt, w->out_w, H, w->out_b, eps, ldeps, STDIFF_F32, STDIFF_EPI_BIAS, nullptr, 0,
               nullptr, 0, stream));
    return STDIFF_OK;
}

// ---------------------------------------------------------------- hipGraph capture / replay
extern "C" int stdiff_graph_begin(void* stream) {
    hipError_t e = hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal);
    return e == hipSuccess ? STDIFF_OK : (int)e;
}

extern "C" int stdiff_graph_end(void* stream, void** graph_exec_out) {
    STDIFF_CHECK_ARG(graph_exec_out);
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture((hipStream_t)stream, &g);
    if (e != hipSuccess) return (int)e;
    if (!g) return STDIFF_ERR_STATE;
    hipGraphExec_t ex = nullptr;
    e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) return (int)e;
    *graph_exec_out = (void*)ex;
    return STDIFF_OK;
}

extern "C" int stdiff_graph_launch(void* graph_exec, void* stream) {
    STDIFF_CHECK_ARG(graph_exec);
    hipError_t e = hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream);
    return e == hipSuccess ? STDIFF_OK : (int)e;
}

extern "C" int stdiff_graph_destroy(void* graph_exec) {
    if (!graph_exec) return STDIFF_OK;
    hipError_t e = hipGraphExecDestroy((hipGraphExec_t)graph_exec);
    return e == hipSuccess ? STDIFF_OK : (int)e;
}
