// libpngpd — the building blocks of the FC stacks (pointnet.py:35-43 and :191-194) as DEVICE FUNCTIONS:
//   fc_tile        one 32x32 output tile of y = x W^T + b (+ epilogue)                 (pngpd_fc_fwd)
//   fc_bwd_tile    one 32x32 tile of dW = g^T x or of dx = g W                         (pngpd_fc_bwd)
//   bn1d_*_tail    BatchNorm1d over the batch for a 32-channel column block, 256 threads
// The one-kernel-per-op entries (pngpd_infer.hip, pngpd_train.hip) and the small-batch column kernels
// (pngpd_head_col.hip: a workgroup owns ALL rows of a 32-channel column, so the BatchNorm behind / in front of the layer
// needs no other workgroup) are both built from these: the same arithmetic in the same order, bit-identical results.
// A "virtual workgroup" = four consecutive waves (the K quarters of one tile); a real workgroup holds VW of them.
#pragma once
#include "pngpd_common.h"

#define FC_RED_FLOATS (3 * 16 * 64)                  // three parked 32x32 partial tiles
#define FC_BWD_LDS_FLOATS (FC_RED_FLOATS + 3 * 32)   // + three parked bias-gradient rows

// BatchNorm1d kernels: block = BN1D_CW channels x BN1D_RL row lanes; lane ry owns rows ry, ry + BN1D_RL, ...
#define BN1D_CW 16
#define BN1D_RL (PNGPD_ASAN ? 16 : 64)   // row lanes; the sanitizer build runs 256-thread workgroups (pngpd_common.h)
#define BN1D_NV 16

// ---------------------------------------------------------------------------------------
// FC forward tile: out = epi(in @ W^T + bias).  One wave = 32 samples x 32 outputs, K contracted with
// v_mfma_f32_32x32x2_f32; A (samples) and B (weight rows) fragments are float4 loads straight from global (both
// operands are small and L2-resident).
// KSPLIT: the workgroup owns ONE tile (rb, cb0) and its four waves each contract a quarter of K (the K = 1024 MFMA
// chain of one wave is 14 us long); partial tiles meet in LDS (`red`, FC_RED_FLOATS) and wave 0 stores.  Contains one
// __syncthreads(); every wave returns.  !KSPLIT: wave w owns tile (rb, 4 cb0 + w), no LDS, no barrier.
// `in` may have been written earlier in the same launch by other workgroups (after a wg_acquire): no __restrict__.
// ---------------------------------------------------------------------------------------
template <bool KSPLIT>
__device__ __forceinline__ void fc_tile(const float *in, int B, int K, const float *__restrict__ W,
                                        const float *__restrict__ bias, int Nout, int epi, float *out, int rb,
                                        int cb0, float *red, bool live = true) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane((tid >> 6) & 3);   // K quarter within the virtual workgroup
    if (KSPLIT && !live) { __syncthreads(); return; }                  // (a virtual workgroup without a tile keeps the barrier)
    const int j = lane & 31, h = lane >> 5;
    const int cb = KSPLIT ? cb0 : cb0 * 4 + wave;
    if (!KSPLIT && cb * 32 >= Nout) return;   // wave-uniform; this variant has no barriers
    int row = rb * 32 + j; row = row < B ? row : B - 1;
    int col = cb * 32 + j; col = col < Nout ? col : Nout - 1;
    const f32x4 *ap = (const f32x4 *)(in + (size_t)row * K) + h;
    const f32x4 *wp = (const f32x4 *)(W + (size_t)col * K) + h;
    f32x16 acc = {0};
    const int KBall = K >> 3;
    const int KB = KSPLIT ? (wave + 1) * (KBall >> 2) : KBall;
    int kb = KSPLIT ? wave * (KBall >> 2) : 0;
    for (; kb + 4 <= KB; kb += 4) {   // 8 loads in flight per lane
        f32x4 a[4], w[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { a[u] = ap[(kb + u) * 2]; w[u] = wp[(kb + u) * 2]; }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc = mfma32(a[u][t], w[u][t], acc);
    }
    for (; kb < KB; ++kb) {
        f32x4 a = ap[kb * 2];
        f32x4 w = wp[kb * 2];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc = mfma32(a[t], w[t], acc);
    }
    if (KSPLIT) {
        if (wave > 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[((wave - 1) * 16 + r) * 64 + lane] = acc[r];
        }
        __syncthreads();
        if (wave > 0) return;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            acc[r] += red[r * 64 + lane] + red[(16 + r) * 64 + lane] + red[(32 + r) * 64 + lane];
    }
    const int c = cb * 32 + j;
    const bool cvalid = c < Nout;
    const float bv = cvalid ? bias[c] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int orow = rb * 32 + mfma_row(r, lane);
        float v = acc[r] + bv;
        if (epi == PNGPD_EPI_RELU) {
            v = (v < 0.f) ? 0.f : v;   // NaN-propagating like F.relu (fmaxf would turn a NaN into 0)
        } else if (epi == PNGPD_EPI_ADD_IDEN3) {
            if (c == 0 || c == 4 || c == 8) v += 1.0f;
        } else if (epi == PNGPD_EPI_LOG_SOFTMAX) {
            float vv = cvalid ? v : -INFINITY;
            float mx = vv;
#pragma unroll
            for (int m = 1; m < 32; m <<= 1) mx = fmaxf(mx, __shfl_xor(mx, m));
            float e = cvalid ? expf(vv - mx) : 0.f;
            float sum = e;
#pragma unroll
            for (int m = 1; m < 32; m <<= 1) sum += __shfl_xor(sum, m);
            v = vv - mx - logf(sum);
        }
        if (cvalid && orow < B) out[(size_t)orow * Nout + c] = v;
    }
}

// ---------------------------------------------------------------------------------------
// Backward of a Linear layer y = x W^T + b (pointnet.py:35-37,191-193), operands read in place (no transposed
// copies): for upstream g (B,Nout)
//   dW (Nout,K) = g^T x   contraction over the batch   — A[i = n][kk = b] = g[b][n], B[kk = b][j = k] = x[b][k]:
//                         both operands are read along rows, i.e. coalesced exactly as the MFMA wants them
//   dx (B,K)    = g W     contraction over Nout        — A[i = b][kk = n] = g[b][n] (float4 along n when
//                         Nout % 8 == 0), B[kk = n][j = k] = W[n][k]
//   db (Nout)   = sum_b g — falls out of the dW tiles of k-block 0 (the A operand IS g)
// One WORKGROUP = one 32x32 output tile (is_w: tile t = nb * kblocks + kbk of dW, else t = rb * kblocks + kbk of dx);
// its four waves each contract a quarter of the reduction range and meet in LDS (a tile's contraction is a chain of up
// to 512 dependent MFMAs: one wave per tile left the launch 4x off the MFMA time with 1.5 workgroups per CU; split four
// ways there are 6 balanced workgroups per CU).  lds: FC_BWD_LDS_FLOATS.  Contains one __syncthreads(); every wave
// returns.  g / x may have been written earlier in the same launch (after a wg_acquire): no __restrict__.
// ---------------------------------------------------------------------------------------
template <bool VEC>
__device__ __forceinline__ void fc_bwd_tile(const float *g, const float *x, const float *__restrict__ W, int B, int K,
                                            int Nout, bool is_w, int t, float *dW, float *dx, float *db, int zero_db,
                                            float *lds, bool live = true) {
    float *red = lds, *dred = lds + FC_RED_FLOATS;
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane((threadIdx.x >> 6) & 3);   // reduction quarter within the virtual workgroup
    if (!live) { __syncthreads(); return; }
    const int kblocks = (K + 31) >> 5;
    f32x16 acc = {0};
    int orow0 = 0, ocol = 0;          // output tile: rows orow0 + mfma_row, column ocol (per lane)
    float dbs = 0.f;
    if (is_w) {
        const int nb = t / kblocks, kbk = t - nb * kblocks;
        const int n = nb * 32 + j, kc = kbk * 32 + j;
        const bool nv = n < Nout, kv = kc < K;
        const float *gp = g + (nv ? n : 0), *xp = x + (kv ? kc : 0);
        const int per = (((B + 3) / 4) + 15) & ~15;               // rows of the batch per wave, a multiple of 16
        const int bb = wave * per, be = (bb + per < B) ? bb + per : B;
        float av[8], bv[8];
        auto fetch = [&](int b0, float (&a)[8], float (&bq)[8]) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int b = b0 + 2 * u + h;
                const bool ok = b < be;
                a[u] = (ok && nv) ? gp[(size_t)b * Nout] : 0.f;
                bq[u] = (ok && kv) ? xp[(size_t)b * K] : 0.f;
            }
        };
        if (bb < be) fetch(bb, av, bv);
        for (int b0 = bb; b0 < be; b0 += 16) {   // group i+1's 16 loads in flight while group i's 8 MFMAs issue
            float na[8], nbv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { na[u] = 0.f; nbv[u] = 0.f; }
            if (b0 + 16 < be) fetch(b0 + 16, na, nbv);
#pragma unroll
            for (int u = 0; u < 8; ++u) { acc = mfma32(av[u], bv[u], acc); dbs += av[u]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) { av[u] = na[u]; bv[u] = nbv[u]; }
        }
        orow0 = nb * 32; ocol = kc;
        dbs += __shfl_xor(dbs, 32);
    } else if (dx) {
        const int rb = t / kblocks, kbk = t - rb * kblocks;
        int row = rb * 32 + j; row = row < B ? row : B - 1;
        const int kc = kbk * 32 + j;
        const bool kv = kc < K;
        const float *wp = W + (kv ? kc : 0);
        const float *gr = g + (size_t)row * Nout;
        if (VEC) {   // Nout % 8 == 0: k-block = 8 values of n, lane (j,h) holds n = 8kb + 4h .. +3
            const f32x4 *ap = (const f32x4 *)gr + h;
            const int KB = Nout >> 3;
            const int per = (((KB + 3) / 4) + 1) & ~1;            // k-blocks per wave, even
            const int k0 = wave * per, k1 = (k0 + per < KB) ? k0 + per : KB;
            f32x4 a[2]; float w[2][4];
            auto fetch = [&](int kb, f32x4 (&aa)[2], float (&ww)[2][4]) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const bool ok = kb + u < k1;
                    const int kk = ok ? kb + u : (k1 > 0 ? k1 - 1 : 0);
                    aa[u] = ok ? ap[kk * 2] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int e = 0; e < 4; ++e) ww[u][e] = (kv && ok) ? wp[(size_t)(kk * 8 + 4 * h + e) * K] : 0.f;
                }
            };
            if (k0 < k1) fetch(k0, a, w);
            for (int kb = k0; kb < k1; kb += 2) {
                f32x4 na[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
                float nw[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
                if (kb + 2 < k1) fetch(kb + 2, na, nw);
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc = mfma32(a[u][e], w[u][e], acc);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    a[u] = na[u];
#pragma unroll
                    for (int e = 0; e < 4; ++e) w[u][e] = nw[u][e];
                }
            }
        } else {
            const int per = (((Nout + 3) / 4) + 7) & ~7;          // values of n per wave, a multiple of 8
            const int n0w = wave * per, n1w = (n0w + per < Nout) ? n0w + per : Nout;
            for (int n0 = n0w; n0 < n1w; n0 += 8) {
                float av[4], bv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int n = n0 + 2 * u + h;
                    const bool ok = n < n1w;
                    av[u] = ok ? gr[n] : 0.f;
                    bv[u] = (ok && kv) ? wp[(size_t)n * K] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) acc = mfma32(av[u], bv[u], acc);
            }
        }
        orow0 = rb * 32; ocol = kc;
    }
    // meet in LDS: waves 1..3 park their tile, wave 0 adds and stores
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((wave - 1) * 16 + r) * 64 + lane] = acc[r];
        if (h == 0) dred[(wave - 1) * 32 + j] = dbs;
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] += red[r * 64 + lane] + red[(16 + r) * 64 + lane] + red[(32 + r) * 64 + lane];
    if (is_w) {
        const int kbk = t % kblocks;
        if (ocol < K) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = orow0 + mfma_row(r, lane);
                if (row < Nout) dW[(size_t)row * K + ocol] = acc[r];
            }
        }
        const int n = orow0 + j;
        // zero_db: the layer feeds a train-mode BatchNorm, whose backward makes sum_b g exactly zero in exact arithmetic
        // (the conv biases of the trunks are treated the same way): write the exact value, not its rounding residue
        if (kbk == 0 && h == 0 && n < Nout) db[n] = zero_db ? 0.f : dbs + dred[j] + dred[32 + j] + dred[64 + j];
    } else if (dx) {
        if (ocol < K) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int orow = orow0 + mfma_row(r, lane);
                if (orow < B) dx[(size_t)orow * K + ocol] = acc[r];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// BatchNorm1d over the batch: the per-element forms every kernel shares
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float bn1d_norm(float z, float mean, float inv, float g, float be, int relu) {
    float v = (z - mean) * inv * g + be;
    if (relu) v = v < 0.f ? 0.f : v;
    return v;
}
__device__ __forceinline__ float bn1d_dz(float gi, float g, float m1, float xh, float m2) {
    return gi * (g - m1 - xh * m2);
}
// nn.BatchNorm1d running statistics: momentum, unbiased variance
__device__ __forceinline__ void bn1d_running(float *rm, float *rv, long long *nbt, int c, float momentum, float mean,
                                             float var, int B) {
    if (rm) {
        rm[c] = (1.f - momentum) * rm[c] + momentum * mean;
        rv[c] = (1.f - momentum) * rv[c] + momentum * var * ((float)B / (float)(B > 1 ? B - 1 : 1));
    }
    if (nbt && c == 0) *nbt += 1;
}
// dlogits = g - exp(logp) * rowsum(g)      (backward of F.log_softmax, pointnet.py:194), one row per thread
__device__ __forceinline__ void log_softmax_bwd_row(const float *g, const float *logp, int K, float *dlogits) {
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += g[k];
    for (int k = 0; k < K; ++k) dlogits[k] = g[k] - expf(logp[k]) * s;
}

// ---------------------------------------------------------------------------------------
// BatchNorm1d of ONE 32-channel column block [c0, c0 + 32) by the first 256 threads of a workgroup (thread = channel cx
// x row group rg of 8; every thread of the workgroup must call — the barriers are the workgroup's), in the arithmetic of bn1d_fwd_train_kernel / bn1d_bwd_kernel: BN1D_RL logical row lanes, lane ry
// accumulates rows ry, ry + BN1D_RL, ... in order, the lanes are then summed in order — each thread here walks
// BN1D_RL / 8 of those lanes.  REG (B <= NR * BN1D_RL): the thread's values are loaded once and stay in registers.  Loads are unconditional on clamped rows + a select (a select around the load itself makes hipcc branch
// and wait per element).  lds: BN1D_RL * 33 floats.  z / dy were written by THIS workgroup before a __syncthreads().
// ---------------------------------------------------------------------------------------
#define BN_TAIL_RPT (BN1D_RL / 8)   // logical row lanes per thread
#define BN_TAIL_NR 4                // REG: rows per logical lane

__device__ __forceinline__ float bn_tail_colsum(float (*red)[33], int cx, int rg, const float (&s)[BN_TAIL_RPT]) {
    __syncthreads();          // previous use of red finished
    if (rg < 8) {             // threads 256.. of a larger workgroup only keep the barriers
#pragma unroll
        for (int q = 0; q < BN_TAIL_RPT; ++q) red[rg * BN_TAIL_RPT + q][cx] = s[q];
    }
    __syncthreads();
    float t = 0.f;
#pragma unroll 8
    for (int i = 0; i < BN1D_RL; ++i) t += red[i][cx];
    return t;
}

template <bool REG, int NR = BN_TAIL_NR>
__device__ __forceinline__ void bn1d_fwd_tail(const float *z, int B, int C, int c0, const float *__restrict__ gamma,
                                              const float *__restrict__ beta, float eps, int relu, float *y,
                                              float *mean_out, float *var_out, float momentum, float *rm, float *rv,
                                              long long *nbt, float *lds) {
    constexpr int RPT = BN_TAIL_RPT;
    float (*red)[33] = (float (*)[33])lds;
    const int cx = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int c = c0 + cx;
    const bool ok = c < C && rg < 8;     // the first 256 threads do the work
    const float *zc = z + (ok ? c : 0);
    const int ry0 = rg * RPT;
    float zv[REG ? RPT * NR : 1];
    float s[RPT];
#pragma unroll
    for (int q = 0; q < RPT; ++q) s[q] = 0.f;
    if (REG) {
#pragma unroll
        for (int i = 0; i < NR; ++i)
#pragma unroll
            for (int q = 0; q < RPT; ++q) {
                const int b = ry0 + q + BN1D_RL * i;
                const float v = zc[(size_t)(b < B ? b : B - 1) * C];
                zv[i * RPT + q] = (ok && b < B) ? v : 0.f;
            }
#pragma unroll
        for (int i = 0; i < NR; ++i)
#pragma unroll
            for (int q = 0; q < RPT; ++q) s[q] += zv[i * RPT + q];
    } else {
        for (int b0 = 0; b0 < B; b0 += BN1D_RL) {
            float v[RPT];
#pragma unroll
            for (int q = 0; q < RPT; ++q) { const int b = b0 + ry0 + q; v[q] = zc[(size_t)(b < B ? b : B - 1) * C]; }
#pragma unroll
            for (int q = 0; q < RPT; ++q) s[q] += (ok && b0 + ry0 + q < B) ? v[q] : 0.f;
        }
    }
    const float mean = bn_tail_colsum(red, cx, rg, s) / (float)B;
#pragma unroll
    for (int q = 0; q < RPT; ++q) s[q] = 0.f;
    if (REG) {
#pragma unroll
        for (int i = 0; i < NR; ++i)
#pragma unroll
            for (int q = 0; q < RPT; ++q) {
                const float d = (ry0 + q + BN1D_RL * i < B) ? zv[i * RPT + q] - mean : 0.f;
                s[q] = fmaf(d, d, s[q]);
            }
    } else {
        for (int b0 = 0; b0 < B; b0 += BN1D_RL) {
            float v[RPT];
#pragma unroll
            for (int q = 0; q < RPT; ++q) { const int b = b0 + ry0 + q; v[q] = zc[(size_t)(b < B ? b : B - 1) * C]; }
#pragma unroll
            for (int q = 0; q < RPT; ++q) {
                const float d = (ok && b0 + ry0 + q < B) ? v[q] - mean : 0.f;
                s[q] = fmaf(d, d, s[q]);
            }
        }
    }
    const float var = bn_tail_colsum(red, cx, rg, s) / (float)B;
    if (!ok) return;
    const float inv = 1.0f / sqrtf(var + eps);
    const float g = gamma[c], be = beta[c];
    float *yc = y + c;
    if (REG) {
#pragma unroll
        for (int i = 0; i < NR; ++i)
#pragma unroll
            for (int q = 0; q < RPT; ++q) {
                const int b = ry0 + q + BN1D_RL * i;
                if (b < B) yc[(size_t)b * C] = bn1d_norm(zv[i * RPT + q], mean, inv, g, be, relu);
            }
    } else {
        for (int b0 = 0; b0 < B; b0 += BN1D_RL) {
            float v[RPT];
#pragma unroll
            for (int q = 0; q < RPT; ++q) { const int b = b0 + ry0 + q; v[q] = zc[(size_t)(b < B ? b : B - 1) * C]; }
#pragma unroll
            for (int q = 0; q < RPT; ++q) {
                const int b = b0 + ry0 + q;
                if (b < B) yc[(size_t)b * C] = bn1d_norm(v[q], mean, inv, g, be, relu);
            }
        }
    }
    if (rg == 0) {
        mean_out[c] = mean; var_out[c] = var;
        bn1d_running(rm, rv, nbt, c, momentum, mean, var, B);
    }
}

// dy: gradient wrt the (post-ReLU if relu) output y.  dz, dgamma, dbeta out.
template <bool REG, int NR = BN_TAIL_NR>
__device__ __forceinline__ void bn1d_bwd_tail(const float *dy, const float *z, const float *y, int B, int C, int c0,
                                              const float *__restrict__ gamma, const float *mean, const float *var,
                                              float eps, int relu, float *dz, float *dgamma, float *dbeta,
                                              float *lds) {
    constexpr int RPT = BN_TAIL_RPT;
    float (*red)[33] = (float (*)[33])lds;
    const int cx = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int c = c0 + cx;
    const bool ok = c < C && rg < 8;     // the first 256 threads do the work
    const int cc = ok ? c : 0;
    const int ry0 = rg * RPT;
    const float mu = mean[cc], inv = 1.0f / sqrtf(var[cc] + eps);
    float gv[REG ? RPT * NR : 1], xh[REG ? RPT * NR : 1];
    float s1[RPT], s2[RPT];
#pragma unroll
    for (int q = 0; q < RPT; ++q) { s1[q] = 0.f; s2[q] = 0.f; }
    auto fetch = [&](int b, float &g, float &x) {     // masked upstream gradient and normalised activation of row b
        const size_t o = (size_t)(b < B ? b : B - 1) * C + cc;
        const float gl = dy[o], zl = z[o], yl = relu ? y[o] : 1.f;
        const bool in = ok && b < B;
        g = (in && yl > 0.f) ? gl : 0.f;
        x = in ? (zl - mu) * inv : 0.f;
    };
    if (REG) {
#pragma unroll
        for (int i = 0; i < NR; ++i)
#pragma unroll
            for (int q = 0; q < RPT; ++q) fetch(ry0 + q + BN1D_RL * i, gv[i * RPT + q], xh[i * RPT + q]);
#pragma unroll
        for (int i = 0; i < NR; ++i)
#pragma unroll
            for (int q = 0; q < RPT; ++q) {
                s1[q] += gv[i * RPT + q];
                s2[q] = fmaf(gv[i * RPT + q], xh[i * RPT + q], s2[q]);
            }
    } else {
        for (int b0 = 0; b0 < B; b0 += BN1D_RL) {
            float g[RPT], x[RPT];
#pragma unroll
            for (int q = 0; q < RPT; ++q) fetch(b0 + ry0 + q, g[q], x[q]);
#pragma unroll
            for (int q = 0; q < RPT; ++q) { s1[q] += g[q]; s2[q] = fmaf(g[q], x[q], s2[q]); }
        }
    }
    const float t1 = bn_tail_colsum(red, cx, rg, s1);
    const float t2 = bn_tail_colsum(red, cx, rg, s2);
    if (!ok) return;
    const float gi = gamma[c] * inv, m1 = t1 / (float)B, m2 = t2 / (float)B;
    if (REG) {
#pragma unroll
        for (int i = 0; i < NR; ++i)
#pragma unroll
            for (int q = 0; q < RPT; ++q) {
                const int b = ry0 + q + BN1D_RL * i;
                if (b < B) dz[(size_t)b * C + c] = bn1d_dz(gi, gv[i * RPT + q], m1, xh[i * RPT + q], m2);
            }
    } else {
        for (int b0 = 0; b0 < B; b0 += BN1D_RL) {
            float g[RPT], x[RPT];
#pragma unroll
            for (int q = 0; q < RPT; ++q) fetch(b0 + ry0 + q, g[q], x[q]);
#pragma unroll
            for (int q = 0; q < RPT; ++q) {
                const int b = b0 + ry0 + q;
                if (b < B) dz[(size_t)b * C + c] = bn1d_dz(gi, g[q], m1, x[q], m2);
            }
        }
    }
    if (rg == 0) { dgamma[c] = t2; dbeta[c] = t1; }
}

