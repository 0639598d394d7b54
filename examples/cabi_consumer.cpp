// A consumer of libpngpd.so that knows nothing about Python or torch: hipMalloc'd buffers, a hipStream_t and the
// entry points of include/pngpd.h.  It scores B synthetic in-gripper clouds with one trunk + one FC layer and
// prints per-output checksums; tests/test_gpu_cabi_consumer.py runs it and compares the numbers with the same
// calls made through the ctypes binding.
//
// build:  hipcc --offload-arch=gfx950 -O2 -Iinclude examples/cabi_consumer.cpp -Lpointnetgpd_amd -lpngpd \
//               -Wl,-rpath,'$ORIGIN/../pointnetgpd_amd' -o examples/cabi_consumer
// run:    examples/cabi_consumer [B] [N]
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
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

int main(int argc, char **argv) {
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
    return 0;
}
