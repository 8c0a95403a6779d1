// A C++ host for normal_train_stDiff's iteration (stDiff_train.py:59-81) over the C ABI: no Python, no torch.  fp32
// compute (the operands are the master weights in place; only the gene-padded in / cond weights are re-packed per step).
//   add-noise + splice -> forward (stdiff_dit_train_forward) -> masked MSE -> backward (stdiff_dit_backward) ->
//   gradient norm -> clip + AdamW, `steps` times on the exported batches; writes <dir>/result.bin = the flat parameters
//   and <dir>/loss.bin.  tests/test_gpu_a_c_host.py compares with TrainEngine.step on the same inputs.
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
static std::map<std::string, void*> dev;

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
        dev[a] = p;
    }
}
static void* T_(const std::string& n) {
    auto it = dev.find(n);
    if (it == dev.end()) { fprintf(stderr, "tensor %s missing\n", n.c_str()); exit(1); }
    return it->second;
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s <export dir>\n", argv[0]); return 1; }
    const std::string dir = argv[1];
    load(dir);
    const int G = (int)scal["G"], Gp = (int)scal["Gp"], H = (int)scal["H"], heads = (int)scal["heads"],
              depth = (int)scal["depth"], C = (int)scal["C"], steps = (int)scal["steps"];
    const long long total = scal["total"];
    const float lr = (float)((double)scal["lr_e12"] * 1e-12);
    if (stdiff_abi_version() != STDIFF_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }
    const int dt = STDIFF_F32;
    hipStream_t st;
    HIPCHECK(hipStreamCreate(&st));

    float* flat_p = (float*)T_("flat_p");
    float* flat_g = (float*)dalloc(total * 4);
    float* m1 = (float*)dalloc(total * 4);
    float* m2 = (float*)dalloc(total * 4);
    void* in_w = dalloc((size_t)H * Gp * 4);        // gene-padded operand copies of in_layer / cond_layer
    void* cond_w = dalloc((size_t)H * Gp * 4);
    auto off = [&](const std::string& n) {
        auto it = scal.find("off." + n);
        if (it == scal.end()) { fprintf(stderr, "offset of %s missing\n", n.c_str()); exit(1); }
        return it->second;
    };
    auto lin = [&](const std::string& n, const void* packed = nullptr) {
        stdiff_train_linear l;
        l.w = packed ? packed : (const void*)(flat_p + off(n + ".weight"));
        l.b = flat_p + off(n + ".bias");
        l.gw = flat_g + off(n + ".weight");
        l.gb = flat_g + off(n + ".bias");
        return l;
    };
    std::vector<stdiff_train_block> blocks(depth);
    for (int k = 0; k < depth; ++k) {
        const std::string p = "blks." + std::to_string(k) + ".";
        blocks[k].qkv = lin(p + "attn.qkv"); blocks[k].proj = lin(p + "attn.proj");
        blocks[k].fc1 = lin(p + "mlp.fc1"); blocks[k].fc2 = lin(p + "mlp.fc2");
        blocks[k].ada = lin(p + "adaLN_modulation.1");
    }
    stdiff_train_weights w = {};
    w.dtype = dt; w.G = G; w.Gp = Gp; w.H = H; w.heads = heads; w.depth = depth;
    w.t0 = lin("time_emb.mlp.0"); w.t2 = lin("time_emb.mlp.2");
    w.cond = lin("cond_layer.0", cond_w); w.in = lin("in_layer.0", in_w);
    w.fada = lin("out_layer.adaLN_modulation.1"); w.out = lin("out_layer.linear");
    w.blocks = blocks.data();
    w.freqs = (const float*)T_("freqs");
    auto repack = [&]() {                     // [H, G] master rows -> [H, Gp] operand rows (pad columns zero)
        SD(stdiff_pack_rows(flat_p + off("in_layer.0.weight"), G, in_w, Gp, dt, H, G, Gp, st));
        SD(stdiff_pack_rows(flat_p + off("cond_layer.0.weight"), G, cond_w, Gp, dt, H, G, Gp, st));
    };
    repack();

    // ---- activations of one batch (everything zero-initialised: operand pad columns must be zero)
    auto f = [&](size_t n) { return dalloc(n * 4); };
    std::vector<stdiff_train_block_acts> ba(depth);
    for (int k = 0; k < depth; ++k) {
        stdiff_train_block_acts& b = ba[k];
        b.h = (float*)f((size_t)C * H); b.hmid = (float*)f((size_t)C * H); b.mod = f((size_t)C * 6 * H);
        b.xn1 = f((size_t)C * H); b.xn2 = f((size_t)C * H); b.qkv = f((size_t)C * 3 * H); b.o = f((size_t)C * H);
        b.lse = (float*)f((size_t)C * heads); b.pa = f((size_t)C * H); b.pm = f((size_t)C * H);
        b.upre = f((size_t)C * 4 * H); b.u = f((size_t)C * 4 * H);
        b.mean1 = (float*)f(C); b.rstd1 = (float*)f(C); b.mean2 = (float*)f(C); b.rstd2 = (float*)f(C);
    }
    const int32_t cu_host[2] = {0, C};
    int32_t* cu = (int32_t*)dalloc(8);
    HIPCHECK(hipMemcpy(cu, cu_host, 8, hipMemcpyHostToDevice));
    stdiff_train_acts a = {};
    a.C = C; a.cu_rows = cu; a.blocks = ba.data();
    a.xp = f((size_t)C * Gp); a.yp = f((size_t)C * Gp); a.e0 = f((size_t)C * 256); a.a1 = f((size_t)C * H); a.e1 = f((size_t)C * H);
    a.temb = (float*)f((size_t)C * H); a.c = (float*)f((size_t)C * H); a.s = f((size_t)C * H);
    a.h_last = (float*)f((size_t)C * H); a.fmod = f((size_t)C * 2 * H); a.xnf = f((size_t)C * H);
    a.meanf = (float*)f(C); a.rstdf = (float*)f(C); a.eps = (float*)f((size_t)C * G);
    a.deps = f((size_t)C * Gp); a.dh = (float*)f((size_t)C * H); a.dxn = f((size_t)C * H);
    a.dmod[0] = f((size_t)C * 6 * H); a.dmod[1] = f((size_t)C * 6 * H); a.dfmod = f((size_t)C * 2 * H);
    a.dbr = f((size_t)C * H); a.du = f((size_t)C * 4 * H); a.dupre = f((size_t)C * 4 * H); a.do_ = f((size_t)C * H);
    a.dqkv = f((size_t)C * 3 * H); a.delta = (float*)f((size_t)C * heads); a.ds32 = (float*)f((size_t)C * H);
    // second set of per-block dY scratch: one grouped weight-gradient launch per block (include/stdiff_hip.h)
    for (int i = 0; i < 3; ++i) a.dbr_b[i] = f((size_t)C * H);
    a.dqkv_b = f((size_t)C * 3 * H);
    a.dc = (float*)f((size_t)C * H); a.dcT = f((size_t)C * H); a.de1 = f((size_t)C * H); a.da1 = f((size_t)C * H);
    a.dhT = f((size_t)C * H);
    const int Cp = (C + 63) / 64 * 64, trows = Gp > H ? Gp : H;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j) a.tr[i][j] = f((size_t)trows * Cp);
    a.splitk_bytes = 64ll << 20;
    a.splitk_ws = dalloc((size_t)a.splitk_bytes);

    float* loss = (float*)dalloc(4);
    float* sumsq = (float*)dalloc(4);
    float* scal4 = (float*)dalloc(16);
    int32_t* step_dev = (int32_t*)dalloc(4);
    std::vector<float> losses(steps);
    for (int i = 0; i < steps; ++i) {
        const std::string s = std::to_string(i);
        const float* x = (const float*)T_("x" + s);
        const float* xc = (const float*)T_("xc" + s);
        const float* noise = (const float*)T_("noise" + s);
        const int64_t* t = (const int64_t*)T_("t" + s);
        const float* mask = (const float*)T_("mask");
        SD(stdiff_add_noise_splice(x, noise, t, (const float*)T_("sa"), (const float*)T_("sb"), mask, nullptr, a.xp, Gp, dt, C, G, st));
        SD(stdiff_pack_rows(xc, G, a.yp, Gp, dt, C, G, Gp, st));
        SD(stdiff_dit_train_forward(&w, &a, t, nullptr, st));
        HIPCHECK(hipMemsetAsync(loss, 0, 4, st));
        SD(stdiff_masked_mse(a.eps, noise, mask, loss, nullptr, a.deps, Gp, dt, 1.0f, C, G, st));
        SD(stdiff_dit_backward(&w, &a, nullptr, nullptr, nullptr, st));
        HIPCHECK(hipMemsetAsync(sumsq, 0, 4, st));
        SD(stdiff_grad_sumsq(flat_g, total, sumsq, st));
        stdiff_adamw_args ad = {};
        ad.params = flat_p; ad.grads = flat_g; ad.exp_avg = m1; ad.exp_avg_sq = m2; ad.n = total;
        ad.sumsq = sumsq; ad.step_dev = step_dev; ad.scal = scal4;
        ad.max_norm = 1.0f; ad.grad_scale = 1.0f; ad.lr = lr; ad.beta1 = 0.9f; ad.beta2 = 0.999f; ad.eps = 1e-8f;
        ad.weight_decay = 0.0f; ad.bf16_mirror = nullptr;
        SD(stdiff_clip_adamw(&ad, st));
        repack();
        HIPCHECK(hipMemcpyAsync(&losses[i], loss, 4, hipMemcpyDeviceToHost, st));
        HIPCHECK(hipStreamSynchronize(st));
    }
    std::vector<float> out(total);
    HIPCHECK(hipMemcpy(out.data(), flat_p, total * 4, hipMemcpyDeviceToHost));
    std::ofstream(dir + "/result.bin", std::ios::binary).write((const char*)out.data(), (std::streamsize)(total * 4));
    std::ofstream(dir + "/loss.bin", std::ios::binary).write((const char*)losses.data(), (std::streamsize)(steps * 4));
    printf("train_host: %d iterations of %d cells, last loss %.6f\n", steps, C, losses[steps - 1]);
    return 0;
}
