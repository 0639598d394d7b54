// libpngpd — inference trunk: fused per-point MLP 3->64->128->1024 + global max-pool.
// Hand-written for gfx950 (MI355X): 64-wide waves, v_mfma_f32_32x32x2_f32 (exact fp32),
// activations staged in LDS, the (B,1024,N) layer-3 activation never touches HBM.
//
// Reference call sites replaced (eval mode, BatchNorm folded by pngpd_fold_conv_bn):
//   PointNetGPD/model/pointnet.py:29-33   STN3d trunk + MaxPool1d          (relu_last = 1)
//   PointNetGPD/model/pointnet.py:140-149 PointNetfeat bmm + trunk + pool  (relu_last = 0, trans)
//
// One workgroup (4 waves) owns cloud b and a contiguous range of 64-point tiles.  Per tile:
//   [xs <- x (x' = x^T T)] | layer 1 VALU -> h1 (LDS) | layer 2 MFMA -> h2 (LDS) |
//   layer 3 MFMA; the max over the tile's points is an in-register reduction (a lane's 16 accumulator
//   registers are 16 points of ONE channel); the per-channel running max over tiles lives in LDS.
// Bias (+ReLU) of layer 3 commute with the max and are applied once at the end.
#include "pngpd_tile.h"

#define TRUNK_LDS_FLOATS (TP * I1S + TP * I2S + 3 * TP + 1024)

// Max of a lane's 32 accumulator values (v_max3 tree) joined with the other row half of the tile.
__device__ __forceinline__ float block_max(const f32x16 &a0, const f32x16 &a1) {
    float t[6];
#pragma unroll
    for (int i = 0; i < 5; ++i) t[i] = max3f(a0[3 * i], a0[3 * i + 1], a0[3 * i + 2]);
    t[5] = a0[15];
    float u[6];
#pragma unroll
    for (int i = 0; i < 5; ++i) u[i] = max3f(a1[3 * i], a1[3 * i + 1], a1[3 * i + 2]);
    u[5] = a1[15];
    const float m = max3f(max3f(t[0], t[1], t[2]), max3f(t[3], t[4], t[5]),
                          max3f(max3f(u[0], u[1], u[2]), max3f(u[3], u[4], u[5]), -INFINITY));
    float lo, hi;
    half_pair(m, lo, hi);
    return fmaxf(lo, hi);
}

__global__ __launch_bounds__(256, 2) void trunk_infer_kernel(
    const float *__restrict__ x, int N, const float *__restrict__ trans,
    const float *__restrict__ w1, const float *__restrict__ b1,
    const float *__restrict__ w2p, const float *__restrict__ b2,
    const float *__restrict__ w3p, const float *__restrict__ b3,
    int relu_last, int T, int S, int CS, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *h1 = smem;              // [TP][I1S] swizzled
    float *h2 = h1 + TP * I1S;     // [TP][I2S] swizzled
    float *xs = h2 + TP * I2S;     // [3][TP]
    float *rm = xs + 3 * TP;       // [1024] running max of the layer-3 pre-bias output
    const Lane L;
    // small launches (latency path) additionally split the 32 channel blocks over CS in {1,2,4} workgroups
    const int cs = blockIdx.x % CS, bs = blockIdx.x / CS;
    const int b = bs / S, s = bs - b * S;
    const int cp0 = cs * (4 / CS), cp1 = cp0 + 4 / CS;   // range of channel-block pairs of this workgroup
    int t0, t1; tile_range(s, S, T, t0, t1);
    const float *xb = x + (size_t)b * 3 * N;
    float tm[9] = {0};
    const bool has_t = trans != nullptr;
    if (has_t) {
#pragma unroll
        for (int i = 0; i < 9; ++i) tm[i] = trans[(size_t)b * 9 + i];
    }
    for (int i = L.tid; i < 1024; i += 256) rm[i] = -INFINITY;
    // A non-finite input coordinate must reach the output like it does through max_pool1d / torch.max (which
    // propagate NaN), but fmaxf drops NaNs and the ReLUs turn them into zeros: remember it and poison this
    // workgroup's pooled row at the end.
    // (+-Inf coordinates poison the row as well: the reference can still produce finite outputs for them — relu(-inf)
    // = 0 — but an infinite point is a corrupt cloud, and NaN is the conservative answer.)
    __shared__ int s_bad;
    static_assert(TP <= 64, "s_bad: the clearing store (tid 0) and the setting stores (tid < TP) must sit in wave 0, "
                            "where stores retire in program order; a larger TP needs a barrier in between");
    if (L.tid == 0) s_bad = 0;
    // layer-3 weight fragments are double-buffered in registers: while channel block ci is on the
    // MFMA pipe the 16 KiB of block ci+1 are in flight from L2 (the last block of a tile prefetches the
    // first block of the next tile, which also covers the tile prologue).
    f32x4 wa[16], wb[16];
    load_wfrag(wa, w3p, L.wave + 8 * cp0, L);
    const L1C l1c = load_l1c(w1, b1, nullptr, nullptr, L);
    // the tile's points are fetched one tile ahead (threads 0..63 hold one point each in registers)
    float px0 = 0.f, px1 = 0.f, px2 = 0.f;
    if (L.tid < TP) {
        int n = t0 * TP + L.tid; n = n < N ? n : N - 1;   // tail: replicate the last point (max unaffected)
        px0 = xb[n]; px1 = xb[N + n]; px2 = xb[2 * N + n];
    }

    for (int tile = t0; tile < t1; ++tile) {
        if (L.tid < TP) {
            float x0 = px0, x1 = px1, x2 = px2;
            if (has_t) {   // x' = x^T @ trans (pointnet.py:140-143)
                x0 = fmaf(px2, tm[6], fmaf(px1, tm[3], px0 * tm[0]));
                x1 = fmaf(px2, tm[7], fmaf(px1, tm[4], px0 * tm[1]));
                x2 = fmaf(px2, tm[8], fmaf(px1, tm[5], px0 * tm[2]));
            }
            xs[L.tid] = x0; xs[TP + L.tid] = x1; xs[2 * TP + L.tid] = x2;
            if (!__builtin_isfinite(px0 + px1 + px2)) s_bad = 1;
            if (tile + 1 < t1) {
                int n = (tile + 1) * TP + L.tid; n = n < N ? n : N - 1;
                px0 = xb[n]; px1 = xb[N + n]; px2 = xb[2 * N + n];
            }
        }
        __syncthreads();
        layer1_tile_swz(xs, l1c, h1, L);   // layer 1 (3 -> 64), VALU: lane = channel, its constants in registers
        __syncthreads();
        {   // layer 2 (64 -> 128), MFMA: wave owns channel block cb = wave, both point blocks
            f32x16 a0, a1;
            const int cb = L.wave;
            f32x4 w2f[8];
            const f32x4 *wp = (const f32x4 *)w2p + (size_t)(cb * 8) * 64 + L.lane;
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) w2f[kb] = wp[kb * 64];
            swz_compute<I1S, 8>(h1, w2f, L, a0, a1);
            const float bias = b2[cb * 32 + L.j];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = mfma_row(r, L.lane);
                h2[swz(row, cb * 32 + L.j, I2S)] = fmaxf(a0[r] + bias, 0.f);
                h2[swz(32 + row, cb * 32 + L.j, I2S)] = fmaxf(a1[r] + bias, 0.f);
            }
        }
        __syncthreads();
#pragma unroll 1
        for (int cp = cp0; cp < cp1; ++cp) {
            const int cbA = L.wave + 8 * cp, cbB = cbA + 4, cbN = L.wave + 8 * (cp + 1 < cp1 ? cp + 1 : cp0);
            f32x16 a0, a1;
            // the block's running max is requested before its 128 MFMAs and merged after them; the two row halves of
            // the tile meet through v_permlane32_swap — no LDS round trip is left between two blocks' matrix streams
            const float rA = rm[cbA * 32 + L.j], rB = rm[cbB * 32 + L.j];
            load_wfrag(wb, w3p, cbB, L);
            swz_compute<I2S, 16>(h2, wa, L, a0, a1);
            float m = block_max(a0, a1);
            if (L.h == 0) rm[cbA * 32 + L.j] = fmaxf(rA, m);
            load_wfrag(wa, w3p, cbN, L);
            swz_compute<I2S, 16>(h2, wb, L, a0, a1);
            m = block_max(a0, a1);
            if (L.h == 0) rm[cbB * 32 + L.j] = fmaxf(rB, m);
        }
        // no barrier here: the next tile's xs/h1 writes do not alias h2, and the barrier before its
        // layer 2 orders the h2 rewrite after every wave's layer-3 reads.
    }
    if (L.h == 0) {   // each (wave, lane<32) reads back exactly the rm entries it wrote
        float *o = out + ((size_t)b * S + s) * 1024;
        for (int ci = 2 * cp0; ci < 2 * cp1; ++ci) {
            const int c = (L.wave + 4 * ci) * 32 + L.j;
            float v = rm[c] + b3[c];
            if (relu_last) v = fmaxf(v, 0.f);
            o[c] = s_bad ? __builtin_nanf("") : v;
        }
    }
}

__global__ void pool_reduce_kernel(const float *__restrict__ part, int S, float *__restrict__ out, int total) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;   // over B*1024
    if (idx >= total) return;
    int b = idx >> 10, c = idx & 1023;
    const float *p = part + (size_t)b * S * 1024 + c;
    float m = p[0];
    bool nan = m != m;
    for (int s = 1; s < S; ++s) { const float v = p[(size_t)s * 1024]; nan |= v != v; m = fmaxf(m, v); }
    out[idx] = nan ? __builtin_nanf("") : m;   // NaN-propagating like torch.max
}

// 4 per CU = two resident rounds of 2.  Measured flat between 1024 and 4096 at B = N = 1024 (round 1); 1024 gives one
// workgroup per cloud there, so the pooled row is written once (4.2 MB instead of 8.4 MB of partial maxima plus a
// second launch to combine them).
#define TRUNK_DEFAULT_TARGET_BLOCKS 1024

// Work decomposition of a launch: S workgroups per cloud along the tiles x CS workgroups along the 32 channel blocks
// (layers 1-2 are recomputed per channel slice: +3 % of a tile's work per extra slice).  512 workgroups are resident
// (2 per CU); a launch whose grid is not a whole number of such rounds ends with CUs running one workgroup alone (one
// wave per SIMD: nothing covers its epilogues) or idle.  Cost model in units of one channel-block pair on one tile:
//   rounds(B S CS) * (ceil(T/S) * (4/CS) * (1 + 0.03 (CS-1)) + 0.6 for the prologue)
// e.g. B = 64, N = 750 (T = 12): S = 12, CS = 1 is 768 workgroups = 1.5 rounds (measured 148 us, 58 % of peak);
// S = 4, CS = 2 is exactly one round of 512 equal workgroups.
static double infer_cost(int B, int T, int S, int CS) {
    const long G = (long)B * S * CS;
    const double w = (double)((T + S - 1) / S) * (4.0 / CS) * (1.0 + 0.03 * (CS - 1));
    return (double)((G + 511) / 512) * (w + 0.6);
}
static int best_cs(int B, int T, int S) {
    int cs = 1;
    double best = infer_cost(B, T, S, 1);
    for (int c = 2; c <= 4; c *= 2) {
        const double v = infer_cost(B, T, S, c);
        if (v < best - 1e-9) { best = v; cs = c; }
    }
    return cs;
}
static int auto_splits(int B, int T) {
    int bestS = 1;
    double best = infer_cost(B, T, 1, best_cs(B, T, 1));
    for (int S = 2; S <= T; ++S) {
        const double v = infer_cost(B, T, S, best_cs(B, T, S));
        if (v < best - 1e-9) { best = v; bestS = S; }
    }
    return bestS;
}
static int resolve_splits(int B, int T, int splits) {
    return (splits > 0) ? (splits > T ? T : splits) : auto_splits(B, T);
}

extern "C" {

int pngpd_trunk_infer_splits(int B, int N, int target_blocks) {
    if (B <= 0 || N <= 0) return 0;
    const int T = (N + TP - 1) / TP;
    return target_blocks > 0 ? pngpd_splits_for(B, T, target_blocks) : auto_splits(B, T);
}

size_t pngpd_trunk_workspace_bytes(int B, int N, int splits) {
    if (B <= 0 || N <= 0) return 0;
    const int S = resolve_splits(B, (N + TP - 1) / TP, splits);
    return S > 1 ? (size_t)B * S * 1024 * sizeof(float) : 0;   // partial maxima of the S workgroups of a cloud
}

int pngpd_trunk_fwd_infer(const float *x, int B, int N, const float *trans,
                          const float *w1, const float *b1, const float *w2p, const float *b2,
                          const float *w3p, const float *b3, int relu_last, int splits,
                          float *out_pool, void *workspace, size_t workspace_bytes, void *stream) {
    if (!x || !w1 || !b1 || !w2p || !b2 || !w3p || !b3 || !out_pool || B <= 0 || N <= 0)
        return PNGPD_ERR_INVALID_ARG;
    const int T = (N + TP - 1) / TP;
    const int S = resolve_splits(B, T, splits);
    float *dst = out_pool;
    if (S > 1) {
        if (!workspace || workspace_bytes < (size_t)B * S * 1024 * sizeof(float)) return PNGPD_ERR_WORKSPACE;
        dst = (float *)workspace;
    }
    const size_t lds = TRUNK_LDS_FLOATS * sizeof(float);
    int st = pngpd_allow_lds((const void *)trunk_infer_kernel, lds);
    if (st != PNGPD_OK) return st;
    const int CS = best_cs(B, T, S);
    hipLaunchKernelGGL(trunk_infer_kernel, dim3((unsigned)B * S * CS), dim3(256), lds, (hipStream_t)stream,
                       x, N, trans, w1, b1, w2p, b2, w3p, b3, relu_last, T, S, CS, dst);
    st = pngpd_launch_status();
    if (st != PNGPD_OK) return st;
    if (S > 1) {
        int total = B * 1024;
        hipLaunchKernelGGL(pool_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                           (const float *)workspace, S, out_pool, total);
        st = pngpd_launch_status();
    }
    return st;
}

}  // extern "C"
