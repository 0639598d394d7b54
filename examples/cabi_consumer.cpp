// A consumer of libpngpd.so that knows nothing about Python or torch: hipMalloc'd buffers, a hipStream_t and the
// entry points of include/pngpd.h.  It scores B synthetic in-gripper clouds with one trunk + one FC layer and
// prints per-output checksums; tests/test_gpu_cabi_consumer.py runs it and compares the numbers with the same
// calls made through the ctypes binding.
//
// build:  hipcc --offload-arch=gfx950 -O2 -Iinclude examples/cabi_consumer.cpp -Lpointnetgpd_amd -lpngpd \
//               -Wl,-rpath,'$ORIGIN/../pointnetgpd_amd' -o examples/cabi_consumer
// run:    examples/cabi_consumer [B] [N] [train]
// With a third argument it also runs ONE training step of a trunk through the fused per-direction entries
// (pngpd_trunk_train_fwd / _bwd with caller-provided save / scratch buffers) followed by pngpd_adam_flat on the
// layer-3 weight, and prints checksums of the pooled output, every gradient and the updated weight.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>

#include "pngpd.h"

#define HIP_OK(e)                                                                        \
    do {                                                                                 \
        hipError_t _e = (e);                                                             \
        if (_e != hipSuccess) { std::fprintf(stderr, "HIP: %s\n", hipGetErrorString(_e)); return 2; } \
    } while (0)
#define PN_OK(e)                                                                         \
    do {                                                                                 \
        int _s = (e);                                                                    \
        if (_s != PNGPD_OK) { std::fprintf(stderr, "pngpd: %s (%d) at line %d\n", pngpd_strerror(_s), _s, __LINE__); return 3; } \
    } while (0)

// The same generator is restated in the Python test: 32-bit LCG -> uniform in [-0.5, 0.5).
struct Lcg {
    uint32_t s;
    float next() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) * (1.0f / 16777216.0f) - 0.5f; }
};

static std::vector<float> fill(Lcg &g, size_t n, float scale, float shift = 0.f) {
    std::vector<float> v(n);
    for (auto &x : v) x = g.next() * scale + shift;
    return v;
}

template <class T>
static int upload(const std::vector<T> &h, T **d) {
    HIP_OK(hipMalloc((void **)d, h.size() * sizeof(T)));
    HIP_OK(hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}

static int run(int argc, char **argv);

// Sanitizer builds leave through _Exit after flushing: ROCm 7.2's ASan runtime CHECK-fails inside the HSA runtime's
// static destructors ("dev_runtime_unloaded_", sanitizer_allocator_device.h) when they free host memory after the
// device allocator is gone — a teardown-order problem of the toolchain that would otherwise turn a clean run into
// exit code 1 and lose the buffered output.  Every kernel has completed (and been checked) by then.
int main(int argc, char **argv) {
    const int rc = run(argc, argv);
    std::fflush(stdout);
    std::fflush(stderr);
#if defined(__SANITIZE_ADDRESS__)
    std::_Exit(rc);
#elif defined(__has_feature)
#if __has_feature(address_sanitizer)
    std::_Exit(rc);
#endif
#endif
    return rc;
}

static int run(int argc, char **argv) {
    const int B = argc > 1 ? std::atoi(argv[1]) : 5, N = argc > 2 ? std::atoi(argv[2]) : 200;
    if (pngpd_abi_version() != PNGPD_ABI_VERSION) { std::fprintf(stderr, "ABI mismatch\n"); return 1; }
    Lcg g{12345u};
    // cloud (B,3,N), per-sample transform (B,3,3) near identity
    auto x = fill(g, (size_t)B * 3 * N, 0.1f);
    auto T = fill(g, (size_t)B * 9, 0.2f);
    for (int b = 0; b < B; ++b) for (int i = 0; i < 3; ++i) T[b * 9 + i * 4] += 1.0f;
    // three conv+BN layers (raw weights, BN statistics) and one FC layer
    const int C[4] = {3, 64, 128, 1024};
    std::vector<float> W[3], bias[3], gam[3], bet[3], mu[3], var[3];
    for (int l = 0; l < 3; ++l) {
        W[l] = fill(g, (size_t)C[l + 1] * C[l], 2.0f / (float)C[l]);
        bias[l] = fill(g, C[l + 1], 0.1f);
        gam[l] = fill(g, C[l + 1], 1.0f, 1.0f);       // [0.5, 1.5)
        bet[l] = fill(g, C[l + 1], 0.2f);
        mu[l] = fill(g, C[l + 1], 0.2f);
        var[l] = fill(g, C[l + 1], 1.0f, 1.0f);
    }
    for (int c = 0; c < 1024; c += 7) gam[2][c] = -gam[2][c];   // negative BN scales on the pooled layer
    auto Wfc = fill(g, (size_t)9 * 1024, 0.05f);
    auto bfc = fill(g, 9, 0.1f);

    hipStream_t st;
    HIP_OK(hipStreamCreate(&st));
    float *dx, *dT, *dW[3], *db[3], *dg[3], *dbe[3], *dmu[3], *dva[3], *dWf[3], *dbf[3], *dWfc, *dbfc, *dpool, *dout;
    if (upload(x, &dx) || upload(T, &dT) || upload(Wfc, &dWfc) || upload(bfc, &dbfc)) return 2;
    for (int l = 0; l < 3; ++l) {
        if (upload(W[l], &dW[l]) || upload(bias[l], &db[l]) || upload(gam[l], &dg[l]) || upload(bet[l], &dbe[l]) ||
            upload(mu[l], &dmu[l]) || upload(var[l], &dva[l])) return 2;
        HIP_OK(hipMalloc((void **)&dWf[l], W[l].size() * sizeof(float)));
        HIP_OK(hipMalloc((void **)&dbf[l], C[l + 1] * sizeof(float)));
        PN_OK(pngpd_fold_conv_bn(dW[l], db[l], dg[l], dbe[l], dmu[l], dva[l], 1e-5f, C[l + 1], C[l],
                                 l == 0 ? PNGPD_LAYOUT_ROWMAJOR : PNGPD_LAYOUT_MFMA_B, dWf[l], dbf[l], st));
    }
    {   // ABI v6: the same three layers folded by ONE launch (pngpd_fold_model) must equal the per-layer entry bit for bit
        pngpd_fold_model_t fm = {};
        float *cW[3], *cb[3];
        fm.n = 3;
        for (int l = 0; l < 3; ++l) {
            HIP_OK(hipMalloc((void **)&cW[l], W[l].size() * sizeof(float)));
            HIP_OK(hipMalloc((void **)&cb[l], C[l + 1] * sizeof(float)));
            pngpd_fold_layer_t &L = fm.layer[l];
            L.W = dW[l]; L.b = db[l]; L.gamma = dg[l]; L.beta = dbe[l]; L.mean = dmu[l]; L.var = dva[l];
            L.eps = 1e-5f; L.C = C[l + 1]; L.K = C[l];
            (l == 0 ? L.row : L.mfma) = cW[l];
            L.bf = cb[l];
        }
        if (pngpd_struct_bytes(2) != sizeof(fm)) { std::fprintf(stderr, "pngpd_fold_model_t size mismatch\n"); return 6; }
        PN_OK(pngpd_fold_model(&fm, st));
        HIP_OK(hipStreamSynchronize(st));
        for (int l = 0; l < 3; ++l) {
            std::vector<float> a(W[l].size()), b2(W[l].size()), c(C[l + 1]), d(C[l + 1]);
            HIP_OK(hipMemcpy(a.data(), dWf[l], a.size() * sizeof(float), hipMemcpyDeviceToHost));
            HIP_OK(hipMemcpy(b2.data(), cW[l], b2.size() * sizeof(float), hipMemcpyDeviceToHost));
            HIP_OK(hipMemcpy(c.data(), dbf[l], c.size() * sizeof(float), hipMemcpyDeviceToHost));
            HIP_OK(hipMemcpy(d.data(), cb[l], d.size() * sizeof(float), hipMemcpyDeviceToHost));
            if (std::memcmp(a.data(), b2.data(), a.size() * sizeof(float)) || std::memcmp(c.data(), d.data(), c.size() * sizeof(float))) {
                std::fprintf(stderr, "pngpd_fold_model != pngpd_fold_conv_bn (layer %d)\n", l); return 6;
            }
            HIP_OK(hipFree(cW[l])); HIP_OK(hipFree(cb[l]));
        }
    }
    HIP_OK(hipMalloc((void **)&dpool, (size_t)B * 1024 * sizeof(float)));
    HIP_OK(hipMalloc((void **)&dout, (size_t)B * 9 * sizeof(float)));
    const int splits = pngpd_trunk_infer_splits(B, N, 0);
    const size_t wsb = pngpd_trunk_workspace_bytes(B, N, splits);
    void *ws = nullptr;
    HIP_OK(hipMalloc(&ws, wsb ? wsb : 4));
    // a too-small workspace must be refused, not overrun
    if (wsb > 4 && pngpd_trunk_fwd_infer(dx, B, N, dT, dWf[0], dbf[0], dWf[1], dbf[1], dWf[2], dbf[2], 0, splits, dpool, ws, 4, st) !=
                       PNGPD_ERR_WORKSPACE) { std::fprintf(stderr, "workspace check missing\n"); return 4; }
    PN_OK(pngpd_trunk_fwd_infer(dx, B, N, dT, dWf[0], dbf[0], dWf[1], dbf[1], dWf[2], dbf[2], 0, splits, dpool, ws, wsb, st));
    PN_OK(pngpd_fc_fwd(dpool, B, 1024, dWfc, dbfc, 9, PNGPD_EPI_ADD_IDEN3, dout, st));
    HIP_OK(hipStreamSynchronize(st));
    std::vector<float> pool((size_t)B * 1024), out((size_t)B * 9);
    HIP_OK(hipMemcpy(pool.data(), dpool, pool.size() * sizeof(float), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(out.data(), dout, out.size() * sizeof(float), hipMemcpyDeviceToHost));
    std::printf("B %d N %d\n", B, N);
    for (int b = 0; b < B; ++b) {
        double s = 0, a = 0;
        for (int c = 0; c < 1024; ++c) { s += pool[(size_t)b * 1024 + c]; a += pool[(size_t)b * 1024 + c] > 0 ? pool[(size_t)b * 1024 + c] : -pool[(size_t)b * 1024 + c]; }
        std::printf("pool %d %.9e %.9e\n", b, s, a);
        std::printf("fc %d", b);
        for (int j = 0; j < 9; ++j) std::printf(" %.9e", out[b * 9 + j]);
        std::printf("\n");
    }
    if (argc <= 3) return 0;

    // ---- one training step of the trunk (train-mode BatchNorm, closed-form backward) through the fused entries
    pngpd_trunk_train_t a = {};
    a.x = dx; a.trans = dT; a.B = B; a.N = N; a.S = pngpd_trunk_splits(B, N, 0);
    a.relu_last = 0; a.precision = 0; a.fp32_side = 0; a.need_bwd = 1; a.eps = 1e-5f; a.momentum = 0.1f;
    a.w1 = dW[0]; a.b1 = db[0]; a.g1 = dg[0]; a.be1 = dbe[0];
    a.w2 = dW[1]; a.b2 = db[1]; a.g2 = dg[1]; a.be2 = dbe[1];
    a.w3 = dW[2]; a.b3 = db[2]; a.g3 = dg[2]; a.be3 = dbe[2];
    a.rm1 = dmu[0]; a.rv1 = dva[0]; a.rm2 = dmu[1]; a.rv2 = dva[1]; a.rm3 = dmu[2]; a.rv3 = dva[2];   // running stats
    a.save_bytes = pngpd_trunk_train_save_bytes(&a);
    a.scratch_bytes = pngpd_trunk_train_scratch_bytes(&a);
    if (!a.save_bytes || !a.scratch_bytes || pngpd_struct_bytes(0) != sizeof(a)) { std::fprintf(stderr, "size query failed\n"); return 5; }
    HIP_OK(hipMalloc(&a.save, a.save_bytes));
    HIP_OK(hipMalloc(&a.scratch, a.scratch_bytes));
    float *dzhat, *ddp, *dgrad, *ddT, *dm, *dv;
    int *didx;
    const size_t goff[13] = {0, 192, 256, 320, 384, 8576, 8704, 8832, 8960, 140032, 141056, 142080, 143104};
    HIP_OK(hipMalloc((void **)&dzhat, (size_t)B * 1024 * sizeof(float)));
    HIP_OK(hipMalloc((void **)&didx, (size_t)B * 1024 * sizeof(int)));
    HIP_OK(hipMalloc((void **)&dgrad, goff[12] * sizeof(float)));
    HIP_OK(hipMalloc((void **)&ddT, (size_t)B * 9 * sizeof(float)));
    a.pooled = dpool; a.idx = didx; a.zhat = dzhat;
    {   // a too-small scratch must be refused
        pngpd_trunk_train_t bad = a; bad.scratch_bytes = 16;
        if (pngpd_trunk_train_fwd(&bad, st) != PNGPD_ERR_WORKSPACE) { std::fprintf(stderr, "scratch check missing\n"); return 4; }
    }
    PN_OK(pngpd_trunk_train_fwd(&a, st));
    auto dpv = fill(g, (size_t)B * 1024, 2.0f);           // upstream gradient dL/dpooled
    if (upload(dpv, &ddp)) return 2;
    a.dp = ddp; a.dT = ddT;
    float **gp[12] = {&a.dW1, &a.db1, &a.dg1, &a.dbe1, &a.dW2, &a.db2, &a.dg2, &a.dbe2, &a.dW3, &a.db3, &a.dg3, &a.dbe3};
    for (int i = 0; i < 12; ++i) *gp[i] = dgrad + goff[i];
    PN_OK(pngpd_trunk_train_bwd(&a, st));
    // Adam (torch defaults, lr 0.005, first step) on the layer-3 weight with its gradient slice
    const size_t n3 = 1024 * 128;
    HIP_OK(hipMalloc((void **)&dm, n3 * sizeof(float))); HIP_OK(hipMalloc((void **)&dv, n3 * sizeof(float)));
    HIP_OK(hipMemsetAsync(dm, 0, n3 * sizeof(float), st)); HIP_OK(hipMemsetAsync(dv, 0, n3 * sizeof(float), st));
    PN_OK(pngpd_adam_flat(dW[2], a.dW3, dm, dv, (long long)n3, 0.005f, nullptr, 0.9f, 0.999f, 1e-8f, 1.0f, nullptr, 1.0f,
                          nullptr, st));
    HIP_OK(hipStreamSynchronize(st));
    auto csum = [&](const float *d, size_t n, const char *name) -> int {
        std::vector<float> h(n);
        HIP_OK(hipMemcpy(h.data(), d, n * sizeof(float), hipMemcpyDeviceToHost));
        double s2 = 0, a2 = 0;
        for (float v : h) { s2 += v; a2 += v > 0 ? v : -v; }
        std::printf("train %s %.9e %.9e\n", name, s2, a2);
        return 0;
    };
    const char *names[12] = {"dW1", "db1", "dg1", "dbe1", "dW2", "db2", "dg2", "dbe2", "dW3", "db3", "dg3", "dbe3"};
    if (csum(dpool, (size_t)B * 1024, "pooled")) return 2;
    for (int i = 0; i < 12; ++i) if (csum(dgrad + goff[i], goff[i + 1] - goff[i], names[i])) return 2;
    if (csum(ddT, (size_t)B * 9, "dT") || csum(dW[2], n3, "W3_after_adam") || csum(dmu[2], 1024, "running_mean3")) return 2;
    return 0;
}
