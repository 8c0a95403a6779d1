// A C++ host for the stDiff reverse loop: no Python, no torch -- hipMalloc'ed buffers and the C ABI of
// include/stdiff_hip.h only.  This is model/sample.py:56-95 (sample_stDiff) as a maintainer binding from C would write
// it: loop-invariant work hoisted (time MLP table, cond_layer(x_cond)), one reverse step captured into a hipGraph,
// replayed T times.  Input: a directory written by stdiff_amd/export.py::export_for_c_host; output: <dir>/result.bin
// (fp32 [N, G]).  tests/test_gpu_a_c_host.py checks it against sample_stDiff(seed=...) bit for bit.
//   build: see __graft_entry__.build()  (hipcc, links stdiff_amd/libstdiff_hip_f16.so or libstdiff_hip.so)
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "stdiff_hip.h"

#define HIPCHECK(e)                                                                          \
    do {                                                                                     \
        hipError_t e_ = (e);                                                                 \
        if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } \
    } while (0)
#define SD(e)                                                                                \
    do {                                                                                     \
        int rc_ = (e);                                                                       \
        if (rc_ != 0) { fprintf(stderr, "%s:%d stdiff status %d: %s\n", __FILE__, __LINE__, rc_, stdiff_error_string(rc_)); exit(3); } \
    } while (0)

static std::map<std::string, long long> scal;
static std::map<std::string, void*> dev;       // device copies of the exported tensors
static std::map<std::string, size_t> bytes;

static void* dalloc(size_t n) {
    void* p = nullptr;
    HIPCHECK(hipMalloc(&p, n ? n : 16));
    HIPCHECK(hipMemset(p, 0, n ? n : 16));
    return p;
}

static void load(const std::string& dir) {
    std::ifstream mf(dir + "/manifest.txt");
    if (!mf) { fprintf(stderr, "no manifest in %s\n", dir.c_str()); exit(1); }
    std::string line;
    while (std::getline(mf, line)) {
        std::istringstream is(line);
        std::string a, b;
        is >> a >> b;
        if (b == "=") { long long v; is >> v; scal[a] = v; continue; }
        const size_t n = std::stoull(b);
        std::vector<char> host(n);
        std::ifstream f(dir + "/" + a + ".bin", std::ios::binary);
        if (!f.read(host.data(), (std::streamsize)n)) { fprintf(stderr, "short file %s\n", a.c_str()); exit(1); }
        void* p = dalloc(n);
        HIPCHECK(hipMemcpy(p, host.data(), n, hipMemcpyHostToDevice));
        dev[a] = p; bytes[a] = n;
    }
}
static void* W(const std::string& n) {
    auto it = dev.find(n);
    if (it == dev.end()) { fprintf(stderr, "tensor %s missing\n", n.c_str()); exit(1); }
    return it->second;
}
static const float* B(const std::string& n) { return (const float*)W(n); }

static int gemm(int M, int N, int K, const void* A, int lda, int dt, const void* Wt, int ldw, const float* b, void* C,
                int ldc, int cdt, int epi, hipStream_t st) {
    stdiff_gemm_args g = {};
    g.M = M; g.N = N; g.K = K; g.A = A; g.lda = lda; g.a_dtype = dt; g.W = Wt; g.ldw = ldw; g.bias = b;
    g.C = C; g.ldc = ldc; g.c_dtype = cdt; g.epi = epi;
    return stdiff_gemm(&g, st);
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s <export dir>\n", argv[0]); return 1; }
    const std::string dir = argv[1];
    load(dir);
    const int G = (int)scal["G"], Gp = (int)scal["Gp"], H = (int)scal["H"], heads = (int)scal["heads"],
              depth = (int)scal["depth"], T = (int)scal["T"], N = (int)scal["N"], chunk = (int)scal["chunk"],
              dt = (int)scal["dtype"];
    const uint64_t seed = (uint64_t)scal["seed"];
    if (stdiff_abi_version() != STDIFF_ABI_VERSION || stdiff_h16_format() != (int)scal["h16"]) {
        fprintf(stderr, "library does not match the export (ABI %d, 16-bit format %d)\n", stdiff_abi_version(), stdiff_h16_format());
        return 1;
    }
    const size_t es = dt == STDIFF_F32 ? 4 : 2;
    hipStream_t st;
    HIPCHECK(hipStreamCreate(&st));

    // ---- weights (NoiseScheduler tables and operand packing were done by the exporter)
    std::vector<stdiff_block_weights> blocks(depth);
    for (int k = 0; k < depth; ++k) {
        const std::string p = "blks." + std::to_string(k) + ".", f = "fused." + std::to_string(k) + ".";
        stdiff_block_weights& b = blocks[k];
        b = {};
        b.qkv_w = W(p + "attn.qkv.weight"); b.qkv_b = B(p + "attn.qkv.bias");
        b.proj_w = W(p + "attn.proj.weight"); b.proj_b = B(p + "attn.proj.bias");
        b.fc1_w = W(p + "mlp.fc1.weight"); b.fc1_b = B(p + "mlp.fc1.bias");
        b.fc2_w = W(p + "mlp.fc2.weight"); b.fc2_b = B(p + "mlp.fc2.bias");
        b.ada_w = W(p + "adaLN_modulation.1.weight"); b.ada_b = B(p + "adaLN_modulation.1.bias");
        if (scal["fused"]) {
            b.ln1_w = W(f + "ln1_w"); b.ln1_b = B(f + "ln1_b"); b.ln2_w = W(f + "ln2_w"); b.ln2_b = B(f + "ln2_b");
            b.gate_w = W(f + "gate_w"); b.gate_b = B(f + "gate_b");
        }
    }
    stdiff_dit_weights w = {};
    w.dtype = dt; w.G = G; w.Gp = Gp; w.H = H; w.heads = heads; w.depth = depth;
    w.in_w = W("in_layer.0.weight"); w.in_b = B("in_layer.0.bias");
    w.out_w = W("out_layer.linear.weight"); w.out_b = B("out_layer.linear.bias");
    w.fada_w = W("out_layer.adaLN_modulation.1.weight"); w.fada_b = B("out_layer.adaLN_modulation.1.bias");
    w.blocks = blocks.data();
    if (scal["fused"]) { w.lnf_w = W("fused.f.ln_w"); w.lnf_b = B("fused.f.ln_b"); }

    // ---- hoisted: time MLP for every timestep (stDiff_model.py:52-85) and cond_layer(x_cond) (:235)
    std::vector<int64_t> tt(T);
    for (int i = 0; i < T; ++i) tt[i] = i;
    int64_t* t_all = (int64_t*)dalloc(T * 8);
    HIPCHECK(hipMemcpy(t_all, tt.data(), T * 8, hipMemcpyHostToDevice));
    void* e0 = dalloc((size_t)T * 256 * es);
    void* e1 = dalloc((size_t)T * H * es);
    float* temb = (float*)dalloc((size_t)T * H * 4);
    SD(stdiff_timestep_embedding(t_all, B("freqs"), e0, 256, dt, T, st));
    SD(gemm(T, H, 256, e0, 256, dt, W("time_emb.mlp.0.weight"), 256, B("time_emb.mlp.0.bias"), e1, H, dt, STDIFF_EPI_SILU, st));
    SD(gemm(T, H, H, e1, H, dt, W("time_emb.mlp.2.weight"), H, B("time_emb.mlp.2.bias"), temb, H, STDIFF_F32, STDIFF_EPI_BIAS, st));
    void* yp = dalloc((size_t)N * Gp * es);
    float* ycond = (float*)dalloc((size_t)N * H * 4);
    SD(stdiff_pack_rows((const float*)W("x_cond"), G, yp, Gp, dt, N, G, Gp, st));
    SD(gemm(N, H, Gp, yp, Gp, dt, W("cond_layer.0.weight"), Gp, B("cond_layer.0.bias"), ycond, H, STDIFF_F32, STDIFF_EPI_BIAS, st));

    // ---- workspace of one row slab = all N rows (chunks of `chunk` rows: attention never crosses a chunk)
    const int n_chunks = (N + chunk - 1) / chunk;
    std::vector<int32_t> cu(n_chunks + 1);
    for (int c = 0; c <= n_chunks; ++c) cu[c] = c * chunk < N ? c * chunk : N;
    int32_t* cu_dev = (int32_t*)dalloc((n_chunks + 1) * 4);
    HIPCHECK(hipMemcpy(cu_dev, cu.data(), (n_chunks + 1) * 4, hipMemcpyHostToDevice));
    stdiff_dit_workspace ws = {};
    ws.R = N; ws.n_chunks = n_chunks; ws.max_chunk_rows = chunk < N ? chunk : N; ws.cu_rows = cu_dev;
    ws.h = (float*)dalloc((size_t)N * H * 4);
    ws.xn = dalloc((size_t)N * H * es); ws.qkv = dalloc((size_t)N * 3 * H * es); ws.o = dalloc((size_t)N * H * es);
    ws.u = dalloc((size_t)N * 4 * H * es);
    if (scal["fused"]) {
        const int parts = H / 256 > 1 ? H / 256 : 1;
        ws.ln_parts = (float*)dalloc((size_t)N * parts * 2 * 4);
        ws.gates = dalloc((size_t)N * 2 * H * es);
    } else {
        ws.mod = dalloc((size_t)N * 6 * H * es); ws.fmod = dalloc((size_t)N * 2 * H * es);
    }
    void* s = dalloc((size_t)N * H * es);
    float* eps = (float*)dalloc((size_t)N * G * 4);

    // ---- x_T (sample.py:56-60), then one reverse step as a graph: silu(c) -> denoiser -> DDPM update + splice -> t -= 1
    float* x = (float*)dalloc((size_t)N * G * 4);
    void* xp = dalloc((size_t)N * Gp * es);                  // the MFMA operand copy of x; its pad columns stay zero
    const float* gt = (const float*)W("gt");
    const float* mask = (const float*)W("mask");
    SD(stdiff_init_sample(x, nullptr, seed, 0, gt, mask, xp, Gp, dt, N, G, st));
    int32_t* t_dev = (int32_t*)dalloc(4);
    const int32_t t_first = T - 1;
    HIPCHECK(hipMemcpyAsync(t_dev, &t_first, 4, hipMemcpyHostToDevice, st));
    HIPCHECK(hipStreamSynchronize(st));
    stdiff_step_args sa = {};
    sa.x = x; sa.eps = eps; sa.gt = gt; sa.mask = mask; sa.coef = (const float*)W("coef");
    sa.t_dev = t_dev; sa.t_first = t_first; sa.z_base = nullptr; sa.seed = seed; sa.row_offset = 0;
    sa.xp = xp; sa.ldxp = Gp; sa.xp_dtype = dt; sa.R = N; sa.G = G; sa.pred_type = 0; sa.z_rows = N;
    void* graph = nullptr;
    SD(stdiff_graph_begin(st));
    SD(stdiff_silu_cond(temb, t_dev, ycond, H, s, H, dt, N, H, st));
    SD(stdiff_dit_forward(&w, &ws, xp, s, eps, G, st));
    SD(stdiff_ddpm_step(&sa, st));
    SD(stdiff_advance_timestep(t_dev, st));
    SD(stdiff_graph_end(st, &graph));
    for (int i = 0; i < T; ++i) SD(stdiff_graph_launch(graph, st));
    HIPCHECK(hipStreamSynchronize(st));
    SD(stdiff_graph_destroy(graph));

    std::vector<float> out((size_t)N * G);
    HIPCHECK(hipMemcpy(out.data(), x, out.size() * 4, hipMemcpyDeviceToHost));
    std::ofstream of(dir + "/result.bin", std::ios::binary);
    of.write((const char*)out.data(), (std::streamsize)(out.size() * 4));
    printf("sample_host: %d spots x %d genes, %d reverse steps, result written\n", N, G, T);
    return 0;
}
